// EXPERIMENT (round 3), NOT part of libnsc_hip.so.  Result: correct (bit-identical to v2, 16/16 block-forward tests), but no
// faster than v2 (0.91 - 1.08x over the codec's shapes, tools/block3_ab.py).  Why, measured with tools/mfma_valu_coexec.hip on
// the same box: a wave that streams fp32 MFMAs back to back is never slowed by its SIMD partner (32.02 cycles per
// v_mfma_f32_16x16x4_f32 in every mix), and the partner's VALU instructions make NO progress meanwhile (v_fma / v_exp loops
// take exactly t_alone + t_mfma): the fp32 matrix instructions run on the vector ALUs.  So a second role can hide LDS / HBM
// latency behind the other role's MFMAs, but none of its elementwise work - the staging masks, tanh epilogue and address
// arithmetic of the P role below simply queue behind the C role's k9 stream (P "stage" 2.4 k cycles alone, 11.4 k beside C:
// tools/fwd3_stamps.py), and the tile takes the sum either way.  To build: copy to nsc_amd/csrc/block3.hip, add it to
// SRCS and call nsc_block_fwd3_dispatch from nsc_gated_block_fwd.
// block3.hip - third generation of the gated-block kernels (nn_core_operator.py:82-112): WAVE-SPECIALISED pipelines.
//
// v2 (block.hip) walks a chain of 64-step tiles with all eight waves in lockstep: every tile is a sequence of phases
// separated by workgroup barriers, and in the elementwise / staging / epilogue part of every phase the matrix pipe idles
// (one 8-wave workgroup owns the CU: there is nobody to overlap with).  v3 keeps the tiles, the chains, the LDS images
// and the per-element instruction sequences of v2, but splits the workgroup into two ROLES that work on CONSECUTIVE
// tiles at the same time:
//     forward   P = waves 0-3: 1x1 conv (h) and both k15 gate convs (g) of tile u + 1        (1100 MFMAs per tile)
//               C = waves 4-7: k9 conv + bias + residual + leaky-relu of tile u               (1260 MFMAs per tile)
// Each SIMD hosts one P wave and one C wave, so its matrix pipe is fed by two INDEPENDENT instruction streams: while
// one wave sits in an epilogue (tanh, LDS / HBM stores), waits on an LDS round trip or stages the next x tile, the other
// one issues MFMAs.  The hand-off (g) is double-buffered in LDS; two workgroup barriers per tile instead of four.
// The C role hands its k9 output (+ bias) to a row-wise copy-out through LDS one interval later: residual x read from
// global memory (L2-hot: the P role staged the same lines one tile earlier) and out written in whole 256-B lines, so
// the x tile in LDS belongs to the P role alone and the output leaves in 16-byte pieces.
#include "block_args.h"
#include <algorithm>

#ifdef NSC_PROBES
// per-round phase stamps of workgroup 0 (s_memtime): slot = 128 * role + 8 * round + k, read back with nsc_probe_read3
__device__ unsigned long long nsc_dbg_stamps3[256];
#define NSC_STAMP3(role, rnd, k)                                                                           \
  do {                                                                                                      \
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && (rnd) < 16) {                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
      nsc_dbg_stamps3[128 * (role) + 8 * (rnd) + (k)] = __builtin_readcyclecounter();                       \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  } while (0)
extern "C" int nsc_probe_read3(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsc_dbg_stamps3), sizeof(unsigned long long) * 256) == hipSuccess ? 0 : -3;
}
#else
#define NSC_STAMP3(role, rnd, k) do { } while (0)
#endif


// ---- software-pipelined MFMA loops (operands of step s + 1 are requested before the MFMAs of step s issue; scheduling
// fences keep the compiler from hoisting dozens of LDS reads ahead of the matrix instructions, which spills) ----
// k9 conv of the C role: NR row tiles (weights in registers) x 4 column tiles, taps [TA, TB).
template <int NR, int NRT, int TA, int TB, int LDG_>
__device__ __forceinline__ void k9_rows(const float (&w9r)[NRT][K9][5], const float* gb, f32x4 (&acc)[NRT][4]) {
  constexpr int NSTEP = (TB - TA) * 5;
  float bb[2][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) bb[0][c] = gb[c * 16 + TA];
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    const int tap = TA + st / 5, k = st % 5;
    if (st + 1 < NSTEP) {
      const int tapn = TA + (st + 1) / 5, kn = (st + 1) % 5;
#pragma unroll
      for (int c = 0; c < 4; ++c) bb[(st + 1) & 1][c] = gb[4 * kn * LDG_ + c * 16 + tapn];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) acc[rr][c] = mfma4(w9r[rr][tap][k], bb[st & 1][c], acc[rr][c]);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// both k15 gate convs of the P role: three row tiles (A from LDS) sharing one B fragment (h), one column tile.  Operands
// are requested TWO steps ahead (three register slots): the P stream shares its SIMD's matrix pipe with a C wave, so one step
// of MFMAs (96 cycles alone) does not cover an LDS round trip.
template <int DIL_, int LDX_, int LDW_>
__device__ __forceinline__ void k15_rows3(const float* ab, const float* hb, f32x4 (&acc)[3]) {
  constexpr int NSTEP = K15 * 5;
  float av[3][3], bv[3];
  auto fetch = [&](int stn, int slot) {
    const int tapn = stn / 5, kn = stn % 5;
    bv[slot] = hb[4 * kn * LDX_ + tapn * DIL_];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) av[slot][rr] = ab[(tapn * NARROW + 4 * kn) * LDW_ + rr * 16];
  };
  fetch(0, 0);
  fetch(1, 1);
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    if (st + 2 < NSTEP) fetch(st + 2, (st + 2) % 3);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) acc[rr] = mfma4(av[st % 3][rr], bv[st % 3], acc[rr]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int RT9, int NK1, int DIL>
__global__ __launch_bounds__(512) void gated_block_fwd3_kernel(BlockArgs a, int ntiles, int tpf) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64, H = 4 + 7 * DIL, WX = TT + 2 * H, WGW = TT + 8, LDX = 112, LDG = 80, CR = 4 * NK1, LDW = 48;
  constexpr int NCT1 = (WX + 15) / 16;      // column tiles of a fresh h tile (7 at dil 2, 6 at dil 1)
  static_assert(NCT1 * 16 <= LDX && 79 + 14 * DIL < NCT1 * 16, "h tile must cover every column the k15 taps read");
  float* xs = sm;                            // [CR][LDX]       x tile of the unit the P role computes next
  float* hs0 = xs + CR * LDX;                // [2][20][LDX]    h, double-buffered (the carried columns are copied across)
  float* gs0 = hs0 + 2 * NARROW * LDX;       // [2][20][LDG]    g, double-buffered: P writes one while C reads the other
  float* w2s = gs0 + 2 * NARROW * LDG;       // [15*20][48]     k15 gate kernels, rows interleaved lin/tanh (block.hip header)
  float* ost = w2s + K15 * NARROW * LDW;     // [C][LDO]        k9 output + bias of the C role, for the row-wise copy-out
  constexpr int LDO = 68;                    // == 4 (mod 32): the D-fragment stores of a wave hit 32 different banks
  const int C = a.C, T = a.T;
  const int Cin = a.Cin;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int first = (int)((long)blockIdx.x * ntiles / gridDim.x), last = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);

  // ---- x tile: prefetch into registers (the four P waves; wave w owns rows 2w + half + 8q), staged into xs one interval
  // later (the C waves keep their registers for the k9 weights and accumulators) ----
  constexpr int NQ4 = (CR + 7) / 8;
  f32x4 pf4[NQ4];
  const __amdgpu_buffer_rsrc_t sx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * Cin * T * 4), 0x00020000);
  const int pi4 = lane & 31, phalf = lane >> 5;
  const int OOB = 0x7ffffff0;
  auto prefetch = [&](int tile) {
    const int tl = __builtin_amdgcn_readfirstlane(tile);
    const int b = tl / tpf, t0 = (tl - b * tpf) * TT;
    const int vo = pi4 < LDX / 4 ? max(((b * Cin + 2 * wave + phalf) * T + t0 - H + 4 * pi4) * 4, 0) : OOB;
#pragma unroll
    for (int q = 0; q < NQ4; ++q)
      pf4[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sx, vo, q * 8 * T * 4, 0));
  };
  auto stage = [&](int tile) {
    const int b = tile / tpf, t0 = (tile - b * tpf) * TT;
    if (pi4 < LDX / 4) {
      const int tb = t0 - H + 4 * pi4;
      const bool m0 = (unsigned)tb < (unsigned)T, m1 = (unsigned)(tb + 1) < (unsigned)T;
      const bool m2 = (unsigned)(tb + 2) < (unsigned)T, m3 = (unsigned)(tb + 3) < (unsigned)T;
      const bool clamped = b == 0 && wave == 0 && phalf == 0 && tb < 0 && tb > -4;
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const int r = 2 * wave + phalf + 8 * q;
        if (r < CR) {
          f32x4 v = pf4[q];
          if (clamped) {     // the offset of the tensor's very first row was clamped to time 0: shift (see block.hip)
            const f32x4 w = v;
            const int sh = -tb;
            v[1] = sh == 1 ? w[0] : 0.f;
            v[2] = sh == 1 ? w[1] : (sh == 2 ? w[0] : 0.f);
            v[3] = sh == 1 ? w[2] : (sh == 2 ? w[1] : w[0]);
          }
          const bool live = r < Cin;
          v[0] = live && m0 ? v[0] : 0.f;
          v[1] = live && m1 ? v[1] : 0.f;
          v[2] = live && m2 ? v[2] : 0.f;
          v[3] = live && m3 ? v[3] : 0.f;
          *reinterpret_cast<f32x4*>(xs + r * LDX + 4 * pi4) = v;
        }
      }
    }
  };
  if (wave < 4) {
    prefetch(first);
    // =============================================== P role ===============================================
    const int pw = wave;
    const int tidp = tid;                    // 0..255
    // The P stream is the sparse one (an LDS round trip per three MFMAs); the C wave on the same SIMD issues eight
    // independent MFMAs per four LDS reads.  At equal priority the matrix pipe alternates between the two and P becomes the
    // critical path (76 cycles per MFMA measured): P wins the arbitration, C fills the gaps.
    __builtin_amdgcn_s_setprio(2);
    // k15 gate kernels -> LDS (P threads only; the C role does not touch w2s)
    {
      constexpr int NE = (K15 * NARROW * LDW + 255) / 256;
      constexpr int HALF = (NE + 1) / 2;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float tmp[HALF];
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const int e = min(tidp + 256 * (i + hh * HALF), K15 * NARROW * LDW - 1);
          const int row = e / LDW, r = e - row * LDW;
          const int ii = r & 15;
          const int c = min((r >> 4) * 8 + (ii >> 2) * 2 + (ii & 1), NARROW - 1);
          const float* src = (ii & 2) ? a.wr : a.wl;
          tmp[i] = src[row * NARROW + c];
        }
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const int e = tidp + 256 * (i + hh * HALF);
          if (e < K15 * NARROW * LDW) w2s[e] = tmp[i];
        }
      }
    }
    const int rt1 = pw >> 1;                 // this wave's row tile of h
    float w1r[NK1];
#pragma unroll
    for (int u = 0; u < NK1; ++u) w1r[u] = a.w1[min(4 * u + kq, Cin - 1) * NARROW + min(rt1 * 16 + l15, NARROW - 1)];
    float b1r[4], blr[3][2], brr[3][2];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) b1r[reg] = a.b1[min(rt1 * 16 + kq * 4 + reg, NARROW - 1)];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = min(r * 8 + kq * 2 + u, NARROW - 1);
        blr[r][u] = a.bl[c];
        brr[r][u] = a.br[c];
      }
    nsc_wait_vmem();
    stage(first);
    if (first + 1 < last) prefetch(first + 1);
    nsc_lds_barrier();                       // beta 0: xs(first) and w2s are in

    for (int r = first - 1; r < last; ++r) {
      const int u = r + 1;
      const bool act = u < last;             // the last round has no unit left for P
      const int b = act ? u / tpf : 0, t0 = act ? (u - b * tpf) * TT : 0;
      const bool fresh = u == first || t0 == 0;
      const bool next_steady = u + 1 < last && (u + 1) - ((u + 1) / tpf) * tpf != 0;
      float* hs = hs0 + (u & 1) * NARROW * LDX;
      float* gs = gs0 + (u & 1) * NARROW * LDG;
      NSC_STAMP3(0, r - first + 1, 0);
      // ---------------- interval a: carried columns, then h = lrelu(W1 x + b1) ----------------
      if (act) {
        if (!fresh) {
          const float* hsp = hs0 + ((u & 1) ^ 1) * NARROW * LDX;
          const float* gsp = gs0 + ((u & 1) ^ 1) * NARROW * LDG;
          if (tidp < NARROW * 8) {
            const int rr = tidp >> 3, cidx = tidp & 7;
            gs[rr * LDG + cidx] = gsp[rr * LDG + cidx + TT];
          }
          for (int e = tidp; e < NARROW * 14 * DIL; e += 256) {
            const int rr = e / (14 * DIL), cidx = 8 + (e - rr * (14 * DIL));
            hs[rr * LDX + cidx] = hsp[rr * LDX + cidx + TT];
          }
        }
        // column tiles of this wave: steady - 2 of the 4 new tiles [2H, 2H + 64); fresh - up to 4 of the NCT1 tiles from 0
        const int npair = fresh ? 2 : 1;
        for (int pr = 0; pr < npair; ++pr) {
          int jb0, jb1;
          if (fresh) {
            jb0 = min((pw & 1) + 4 * pr, NCT1 - 1) * 16;
            jb1 = min((pw & 1) + 4 * pr + 2, NCT1 - 1) * 16;
          } else {
            jb0 = 2 * H + (pw & 1) * 32;
            jb1 = jb0 + 16;
          }
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          const float* xc0 = xs + kq * LDX + jb0 + l15;
          const float* xc1 = xs + kq * LDX + jb1 + l15;
#pragma unroll
          for (int k = 0; k < NK1; ++k) {
            acc0 = mfma4(w1r[k], xc0[4 * k * LDX], acc0);
            acc1 = mfma4(w1r[k], xc1[4 * k * LDX], acc1);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = (e ? jb1 : jb0) + l15;
            const int t = t0 - H + j;
            const bool live = j < WX && t >= 0 && t < T;
            const bool save = a.h_out && live && j >= (fresh ? H : 2 * H) && j < (next_steady ? WX : H + TT);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int o = rt1 * 16 + kq * 4 + reg;
              if (o < NARROW) {
                float v = (e ? acc1[reg] : acc0[reg]) + b1r[reg];
                v = v > 0.f ? v : NSC_LRELU_ALPHA * v;
                hs[o * LDX + j] = live ? v : 0.f;
                if (save) a.h_out[((long)b * NARROW + o) * T + t] = v;
              }
            }
          }
        }
      }
      NSC_STAMP3(0, r - first + 1, 1);
      nsc_lds_barrier();                     // beta 1
      NSC_STAMP3(0, r - first + 1, 2);
      // ---------------- interval b: next x tile -> LDS, then both k15 gate convs ----------------
      if (r + 2 < last) {
        stage(r + 2);
        NSC_STAMP3(0, r - first + 1, 7);
        if (r + 3 < last) prefetch(r + 3);
      }
      NSC_STAMP3(0, r - first + 1, 3);
      if (act) {
        const int joff = fresh ? 0 : 8;
        // job set 1: column tile pw (from joff), all three row tiles - the B fragment (h) is shared by the three MFMAs
        {
          f32x4 acc[3];
          acc[0] = acc[1] = acc[2] = (f32x4){0.f, 0.f, 0.f, 0.f};
          const float* ab = w2s + kq * LDW + l15;
          const float* hb = hs + kq * LDX + pw * 16 + l15 + joff;
          k15_rows3<DIL, LDX, LDW>(ab, hb, acc);
          NSC_STAMP3(0, r - first + 1, 4);
          const int jj = pw * 16 + l15 + joff;
          const int t = t0 - 4 + jj;
          const bool live = jj < WGW && t >= 0 && t < T;
          const bool save = a.lin_out && live && jj >= (fresh ? 4 : 8) && jj < (next_steady ? WGW : 4 + TT);
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int c = rr * 8 + kq * 2 + k;
              if (c < NARROW) {
                const float lin = acc[rr][k] + blr[rr][k];
                const float th = tanhf(acc[rr][2 + k] + brr[rr][k]);
                gs[c * LDG + jj] = live ? lin * th : 0.f;
                if (save) {
                  const long gi = ((long)b * NARROW + c) * T + t;
                  a.lin_out[gi] = lin;
                  a.th_out[gi] = th;
                  a.g_out[gi] = lin * th;
                }
              }
            }
        }
        // job set 2 (fresh tiles only): the fifth column tile [64, 80), one row tile each on waves 1..3
        if (fresh && pw > 0) {
          const int rr = pw - 1;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const float* ab = w2s + kq * LDW + rr * 16 + l15;
          const float* hb = hs + kq * LDX + 64 + l15;
#pragma unroll
          for (int tap = 0; tap < K15; ++tap)
#pragma unroll
            for (int k = 0; k < 5; ++k) acc = mfma4(ab[(tap * NARROW + 4 * k) * LDW], hb[4 * k * LDX + tap * DIL], acc);
          const int jj = 64 + l15;
          const int t = t0 - 4 + jj;
          const bool live = jj < WGW && t >= 0 && t < T;
          const bool save = a.lin_out && live && jj < (next_steady ? WGW : 4 + TT);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = rr * 8 + kq * 2 + k;
            if (c < NARROW) {
              const float lin = acc[k] + a.bl[c];
              const float th = tanhf(acc[2 + k] + a.br[c]);
              gs[c * LDG + jj] = live ? lin * th : 0.f;
              if (save) {
                const long gi = ((long)b * NARROW + c) * T + t;
                a.lin_out[gi] = lin;
                a.th_out[gi] = th;
                a.g_out[gi] = lin * th;
              }
            }
          }
        }
      }
      NSC_STAMP3(0, r - first + 1, 5);
      nsc_lds_barrier();                     // beta 2: g(u) complete; C is done with g(u - 1)
      NSC_STAMP3(0, r - first + 1, 6);
    }
  } else {
    // =============================================== C role ===============================================
    const int cw = wave - 4;
    constexpr int NRT = RT9 > 4 ? 2 : 1;     // row tiles per wave: {cw, cw + 4}
    const bool has2 = NRT == 2 && cw + 4 < RT9;          // wave-uniform
    float w9r[NRT][K9][5], b9r[NRT][4];
#pragma unroll
    for (int rr = 0; rr < NRT; ++rr) {
      const int rt = min(cw + 4 * rr, RT9 - 1);
#pragma unroll
      for (int tap = 0; tap < K9; ++tap)
#pragma unroll
        for (int k = 0; k < 5; ++k) w9r[rr][tap][k] = a.w9[(tap * NARROW + 4 * k + kq) * C + min(rt * 16 + l15, C - 1)];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) b9r[rr][reg] = a.b9[min(rt * 16 + kq * 4 + reg, C - 1)];
    }
    const __amdgpu_buffer_rsrc_t so =
        __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((long)a.B * C * T * 4), 0x00020000);
    const int tidc = tid - 256;
    f32x4 acc[NRT][4];
    // Row-wise copy-out of a finished tile: out = act(ost + x) in whole 256-B lines (16-byte loads of the residual x and
    // 16-byte stores), one interval AFTER the k9 conv wrote ost, beside the P role's MFMAs.  The D-fragment form (a dword per
    // lane, 64 memory instructions per wave and tile) overflowed the wave's memory queue and was the critical path.
    constexpr int NROWQ = (RT9 * 16 * 16 + 255) / 256;
    auto copy_out = [&](int tile) {
      const int b = tile / tpf, t0 = (tile - b * tpf) * TT;
      f32x4 xr[NROWQ];
#pragma unroll
      for (int j = 0; j < NROWQ; ++j) {
        const int id = tidc + 256 * j, row = id >> 4, q4 = id & 15;
        const bool ok = row < C && t0 + 4 * q4 < T;
        const int vo = NK1 == 1 ? (b * T + t0 + 4 * q4) * 4 : ((b * C + row) * T + t0 + 4 * q4) * 4;
        xr[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sx, ok ? vo : OOB, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < NROWQ; ++j) {
        const int id = tidc + 256 * j, row = id >> 4, q4 = id & 15;
        const int t = t0 + 4 * q4;
        if (row < C && t < T) {
          f32x4 v = *reinterpret_cast<const f32x4*>(ost + row * LDO + 4 * q4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // a row of x may end inside this 16-byte piece (T not a multiple of 4): the load then ran into the next row
            float y = v[e] + (t + e < T ? xr[j][e] : 0.f);
            if (!a.flat) y = y > 0.f ? y : NSC_LRELU_ALPHA * y;
            v[e] = y;
          }
          const int vo = ((b * C + row) * T + t) * 4;
          if (t + 3 < T) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) int, v), so, vo, 0, 0);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (t + e < T) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[e]), so, vo + 4 * e, 0, 0);
          }
        }
      }
    };
    constexpr int TSPLIT = 2;                // taps of the k9 conv that run in interval a (beside the P role's 1x1)
    nsc_wait_vmem();
    nsc_lds_barrier();                       // beta 0

    for (int r = first - 1; r < last; ++r) {
      const int i = r;
      const bool act = i >= first;           // the first round has no tile for C yet
      const float* gb = gs0 + (i & 1) * NARROW * LDG + kq * LDG + l15;
      NSC_STAMP3(1, r - first + 1, 0);
      // ---------------- interval a: copy-out of the previous tile, first taps of the k9 conv on g(i) ----------------
      if (i > first) copy_out(i - 1);
      NSC_STAMP3(1, r - first + 1, 1);
      if (act) {
#pragma unroll
        for (int rr = 0; rr < NRT; ++rr)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[rr][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has2) k9_rows<NRT, NRT, 0, TSPLIT, LDG>(w9r, gb, acc);
        else k9_rows<1, NRT, 0, TSPLIT, LDG>(w9r, gb, acc);
      }
      NSC_STAMP3(1, r - first + 1, 2);
      nsc_lds_barrier();                     // beta 1
      NSC_STAMP3(1, r - first + 1, 3);
      // ---------------- interval b: remaining taps, k9 output + bias -> ost ----------------
      NSC_STAMP3(1, r - first + 1, 4);
      if (act) {
        if (has2) k9_rows<NRT, NRT, TSPLIT, K9, LDG>(w9r, gb, acc);
        else k9_rows<1, NRT, TSPLIT, K9, LDG>(w9r, gb, acc);
        NSC_STAMP3(1, r - first + 1, 5);
#pragma unroll
        for (int rr = 0; rr < NRT; ++rr) {
          if (rr == 1 && !has2) break;
          const int ob = 16 * (cw + 4 * rr) + 4 * kq;
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              if (ob + reg < C) ost[(ob + reg) * LDO + 16 * c + l15] = acc[rr][c][reg] + b9r[rr][reg];
        }
      }
      NSC_STAMP3(1, r - first + 1, 6);
      nsc_lds_barrier();                     // beta 2
    }
    copy_out(last - 1);
  }
}

template <int RT9, int NK1, int DIL>
static int launch_block_fwd3(const BlockArgs& a, hipStream_t st) {
  constexpr int CR = 4 * NK1;
  const size_t smem = ((size_t)(CR + 2 * NARROW) * 112 + (size_t)2 * NARROW * 80 + (size_t)K15 * NARROW * 48 + (size_t)a.C * 68) * sizeof(float);
  auto kern = gated_block_fwd3_kernel<RT9, NK1, DIL>;
  static const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd3: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, 64);
  const int ntiles = a.B * tpf;
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, 256)), dim3(512), smem, st, a, ntiles, tpf);
  NSC_CHECK_LAUNCH("gated_block_fwd3");
  return NSC_OK;
}

// Dispatch for the shapes the codec uses; returns NSC_ERR_UNSUPPORTED for anything else (the caller falls back to v2 / v1).
int nsc_block_fwd3_dispatch(const BlockArgs& a, hipStream_t st) {
  if ((long)a.B * a.C * a.T * 4 >= (1L << 31)) return NSC_ERR_UNSUPPORTED;      // 32-bit buffer offsets
  const int C = a.C, dil = a.dil;
  if (a.Cin == 1) {
    if (C == 100) return dil == 1 ? launch_block_fwd3<7, 1, 1>(a, st) : launch_block_fwd3<7, 1, 2>(a, st);
    if (C == 50) return dil == 1 ? launch_block_fwd3<4, 1, 1>(a, st) : launch_block_fwd3<4, 1, 2>(a, st);
    return NSC_ERR_UNSUPPORTED;
  }
  if (C == 100) return dil == 1 ? launch_block_fwd3<7, 25, 1>(a, st) : launch_block_fwd3<7, 25, 2>(a, st);
  if (C == 50) return dil == 1 ? launch_block_fwd3<4, 13, 1>(a, st) : launch_block_fwd3<4, 13, 2>(a, st);
  return NSC_ERR_UNSUPPORTED;
}

int nsc_block_dgrad3_dispatch(const BlockDgradArgs& a, int cin1, hipStream_t st) { return NSC_ERR_UNSUPPORTED; }   // (below)
