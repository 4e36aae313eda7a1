// block.hip - the whole gated bottleneck block of the reference (nn_core_operator.py:82-112) as ONE kernel:
//     h = lrelu(W1 x + b1)                      1x1,  C  -> 20
//     g = (Wl *d h + bl) . tanh(Wr *d h + br)    k15 dilated, 20 -> 20 (two weight sets), GLU gate
//     y = W9 * g + b9 + x ; out = lrelu(y) | y   k9,   20 -> C, residual, optional activation
// The 20-channel intermediates never leave the CU: x tile (with a 4+7d halo) is staged once in LDS, h and g
// live in LDS, the residual is re-read from the staged x tile.  HBM traffic per block = read x (x1.5 halo at
// a 64-step tile) + write out, instead of ~14 tensor passes in the unfused path.  All three convs run on
// v_mfma_f32_16x16x4_f32 (exact fp32).  The two gate convs share one MFMA pass: their output rows are
// interleaved (rows 4q+{0,1} = linear branch of channels 2q,2q+1; rows 4q+{2,3} = tanh branch of the same
// channels) so that the D fragment of a lane holds both branches of a channel and the gate is lane-local.
#include "nsc_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "block_args.h"
#include "block_common.h"

#ifdef NSC_PROBES
extern "C" int nsc_probe_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsc_dbg_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3;
}
#endif





// TT = 64 output steps per workgroup, 8 waves (two per SIMD at one workgroup, four at the usual two workgroups per CU):
// every phase is split into (row tile, column tile) jobs dealt over the 8 waves, so each wave's MFMA chain is short
// and its partner waves hide the LDS / weight-fetch latency.
template <int RT9>
__global__ __launch_bounds__(512) void gated_block_fwd_kernel(BlockArgs a, int ldx, int ldg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64;
  const int C = a.C, T = a.T, d = a.dil;
  const int H = 4 + 7 * d;
  const int WX = TT + 2 * H;          // staged x / h width
  const int WG = TT + 8;              // g width
  const int C4 = (C + 3) & ~3;
  float* xs = sm;                     // [C4][ldx]
  float* hs = xs + C4 * ldx;          // [NARROW][ldx]
  float* gs = hs + NARROW * ldx;      // [NARROW][ldg]  (aliasing g over the dead x tile -> 3 workgroups/CU measured no faster)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7, provably wave-uniform
  const int l15 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;

  // ---- phase 0: stage x tile (zero outside [0,T), zero pad rows) ----
  nsc_stage_rows<8>(xs, ldx, C4, C, WX, a.x + (long)b * C * T, T, t0 - H, T, 0, wave, lane);
  __syncthreads();

  // ---- phase 1: h = lrelu(W1 x + b1) on all WX columns; column tile ct -> wave ct (7 tiles at d=2) ----
  {
    const int nct = (WX + 15) >> 4;
    const int ncq = C4 >> 2;
    for (int ct = wave; ct < nct; ct += 8) {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const int j = ct * 16 + l15;
      const float* xcol = xs + j;
      constexpr int G1 = 8;
      float an0[G1], an1[G1];
      const int o1c = 16 + l15 < NARROW ? 16 + l15 : NARROW - 1;   // rows >= 20 of the second tile: clamped, never stored
      auto fetch1 = [&](int cq0) {
#pragma unroll
        for (int u = 0; u < G1; ++u) {
          const int ci = (cq0 + u) * 4 + kq;
          const bool ok = (cq0 + u) < ncq && ci < C;
          an0[u] = nsc_ldm(a.w1, ci * NARROW + l15, ok);
          an1[u] = nsc_ldm(a.w1, ci * NARROW + o1c, ok);
        }
      };
      fetch1(0);
      for (int cq0 = 0; cq0 < ncq; cq0 += G1) {      // padded to a multiple of G1: masked weights are zero
        float c0[G1], c1[G1];
#pragma unroll
        for (int u = 0; u < G1; ++u) { c0[u] = an0[u]; c1[u] = an1[u]; }
        fetch1(cq0 + G1);
#pragma unroll
        for (int u = 0; u < G1; ++u) {
          const int cqc = (cq0 + u < ncq) ? cq0 + u : 0;
          const float bv = xcol[(cqc * 4 + kq) * ldx];
          acc0 = mfma4(c0[u], bv, acc0);
          acc1 = mfma4(c1[u], bv, acc1);
        }
      }
      const int t = t0 - H + j;
      const bool live = j < WX && t >= 0 && t < T;   // h outside the frame is ZERO padding of the k15 convs
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int o0 = kq * 4 + reg, o1 = 16 + kq * 4 + reg;
        float v0 = acc0[reg] + a.b1[o0];
        v0 = v0 > 0.f ? v0 : NSC_LRELU_ALPHA * v0;
        hs[o0 * ldx + j] = live ? v0 : 0.f;
        float v1 = 0.f;
        if (o1 < NARROW) {
          v1 = acc1[reg] + a.b1[o1];
          v1 = v1 > 0.f ? v1 : NSC_LRELU_ALPHA * v1;
          hs[o1 * ldx + j] = live ? v1 : 0.f;
        }
        if (a.h_out && live && j >= H && j < H + TT) {
          a.h_out[((long)b * NARROW + o0) * T + t] = v0;
          if (o1 < NARROW) a.h_out[((long)b * NARROW + o1) * T + t] = v1;
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: both k15 dilated gate convs in one MFMA pass, gate in registers, g -> LDS ----
  // g column jj <-> frame time t0 - 4 + jj ; reads h column jj + tap*d.  15 jobs = 3 row tiles x 5 column tiles (80 >= 72);
  // job q -> (row tile q % 3, column tile q / 3); wave w runs jobs w and w + 8 (wave 7's second job is a discarded duplicate).
  {
    const int i = l15;
    const float* wsel = (i & 2) ? a.wr : a.wl;
    int jrt[2], jcol[2], crow[2];
    bool jlive[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int q = wave + 8 * e;
      jlive[e] = q < 15;
      const int qq = jlive[e] ? q : wave;
      jrt[e] = qq % 3;
      jcol[e] = (qq / 3) * 16 + l15;
      const int c = jrt[e] * 8 + (i >> 2) * 2 + (i & 1);
      crow[e] = c < NARROW ? c : NARROW - 1;          // channels 20..23: clamped, never stored
    }
    f32x4 acc[2];
    acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int G2 = 5;                               // one tap (5 k-steps of 4 channels) per group
    float an[G2][2];
    auto fetch2 = [&](int tapf) {
      const int tf = tapf < K15 ? tapf : K15 - 1;       // past the end: re-fetch the last tap (unused)
#pragma unroll
      for (int u = 0; u < G2; ++u) {
        const int rowb = (tf * NARROW + u * 4 + kq) * NARROW;
        an[u][0] = wsel[rowb + crow[0]];
        an[u][1] = wsel[rowb + crow[1]];
      }
    };
    fetch2(0);
    for (int tap = 0; tap < K15; ++tap) {
      float ac[G2][2];
#pragma unroll
      for (int u = 0; u < G2; ++u) { ac[u][0] = an[u][0]; ac[u][1] = an[u][1]; }
      fetch2(tap + 1);
#pragma unroll
      for (int u = 0; u < G2; ++u) {
        const float* hrow = hs + (u * 4 + kq) * ldx + tap * d;
        acc[0] = mfma4(ac[u][0], hrow[jcol[0]], acc[0]);
        acc[1] = mfma4(ac[u][1], hrow[jcol[1]], acc[1]);
      }
    }
    // epilogue: lane holds (lin c0, lin c1, gate c0, gate c1) of one time step
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (!jlive[e]) continue;
      const int jj = jcol[e];
      const int c0 = jrt[e] * 8 + kq * 2;
      const int t = t0 - 4 + jj;
      const bool live = jj < WG && t >= 0 && t < T;   // g outside the frame is ZERO padding of the k9 conv
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = c0 + u;
        if (c < NARROW && jj < ldg) {
          const float lin = acc[e][u] + a.bl[c];
          const float th = nsc_tanh(acc[e][2 + u] + a.br[c]);
          gs[c * ldg + jj] = live ? lin * th : 0.f;
          if (a.lin_out && live && jj >= 4 && jj < 4 + TT) {
            const long gi = ((long)b * NARROW + c) * T + t;
            a.lin_out[gi] = lin;
            a.th_out[gi] = th;
            a.g_out[gi] = lin * th;
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 3: y = W9 * g + b9 + x ; wave w owns output columns [16 (w&3), +16) and row tiles [(w>>2) RH, +RH) ----
  {
    constexpr int RH = (RT9 + 1) / 2;
    const int rbase = (wave >> 2) * RH;
    f32x4 acc[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tt = (wave & 3) * 16 + l15;
    int ocl[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) { const int o = (rbase + r) * 16 + l15; ocl[r] = o < C ? o : C - 1; }
    constexpr int G3 = 5;                       // one tap per group
    float an[G3][RH];
    auto fetch3 = [&](int tapf) {
      const int tf = tapf < K9 ? tapf : K9 - 1;
#pragma unroll
      for (int u = 0; u < G3; ++u) {
        const int rowb = (tf * NARROW + u * 4 + kq) * C;
#pragma unroll
        for (int r = 0; r < RH; ++r) an[u][r] = a.w9[rowb + ocl[r]];
      }
    };
    fetch3(0);
    for (int tap = 0; tap < K9; ++tap) {
      float ac[G3][RH];
#pragma unroll
      for (int u = 0; u < G3; ++u)
#pragma unroll
        for (int r = 0; r < RH; ++r) ac[u][r] = an[u][r];
      fetch3(tap + 1);
#pragma unroll
      for (int u = 0; u < G3; ++u) {
        const float bv = gs[(u * 4 + kq) * ldg + tt + tap];
#pragma unroll
        for (int r = 0; r < RH; ++r) acc[r] = mfma4(ac[u][r], bv, acc[r]);
      }
    }
    const int t = t0 + tt;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < RH; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int o = (rbase + r) * 16 + kq * 4 + reg;
          if (o < C) {
            float v = acc[r][reg] + a.b9[o] + xs[o * ldx + H + tt];
            if (!a.flat) v = v > 0.f ? v : NSC_LRELU_ALPHA * v;
            a.out[((long)b * C + o) * T + t] = v;
          }
        }
    }
  }
}

// -----------------------------------------------------------------------------------------------------
// v2: the same block, PERSISTENT and weight-stationary.  One workgroup per CU walks (frame, 64-step tile) pairs.
// v1 fetched every MFMA's A fragment (a weight) from global memory - one vmem instruction, three address VALU ops
// and a wait per MFMA, used exactly once per workgroup - which held it at ~27 % of the fp32 MFMA rate.  Here the
// weights are loaded ONCE per workgroup: the 1x1 (50 fragments/lane) and this wave's k9 row tile (45) stay in
// registers for the life of the kernel, the two k15 gate kernels (row-interleaved as in v1) sit in LDS, and the inner
// loops are {ds_read with an immediate offset, v_mfma} only.  The next tile's x is prefetched into registers
// while the current tile computes.  Built for the shapes the codec uses (C = 4*NK1 rounded, dil 1|2); other shapes
// take v1.
// -----------------------------------------------------------------------------------------------------
// ---- neighbour flags of a PAIR launch (one launch = two blocks of a stack, the second reads what the first wrote) ----
// The second block's tiles of workgroup w need columns its neighbours w - 1 / w + 1 produced in the first block (the halo of
// its first / last tile); every other input of those tiles is the workgroup's own.  So instead of a kernel boundary - all 256
// workgroups drained, a dispatch, and every workgroup's weight prologue and first tile exposed again - a workgroup publishes
// "my first-block tiles are in memory" (agent-scope release: its L2 is written back) and, before the first x / dy load of the
// second block, waits for its two neighbours' flags (agent-scope acquire: stale lines invalidated).  All workgroups of the launch
// are co-resident (grid <= CUs, one workgroup per CU by LDS: the launcher checks), so the wait cannot deadlock; it is bounded
// anyway and counts a time-out in flags[gridDim.x] (results are then wrong; the engine checks the counter).
// Flags are zeroed by the caller before the launch (the engine: in the step's opening launch).
// Memory model (gfx950, one L2 per XCD, the L2s not coherent with each other).  An agent-scope FENCE pair writes back and
// INVALIDATES a whole L2 per workgroup: measured, every pair launch 55 us slower than its two blocks launched one by one (the 32
// workgroups of an XCD keep emptying the L2 the others work from).  A write-back alone (buffer_wbl2 sc1 per wave, no invalidation)
// still cost 15-20 us per pair launch (timing builds EXP=64 / 128: profiles/r04b_pair_launch_experiments.txt).  So only what
// crosses workgroups is made coherent, operation by operation:
//   * the first block's tensor that the second block reads leaves through WRITE-THROUGH stores (sc0 sc1: in memory once acknowledged);
//   * publish: every wave waits for its stores (vmcnt), barrier, one relaxed agent-scope store (sc1) sets the flag;
//   * wait: relaxed agent-scope loads of the flag (sc1: from memory), no fence;
//   * the second block reads that tensor with sc0 sc1 loads (served from memory / the memory-side Infinity Cache, never from an
//     L1 / L2 line) - its only input that another workgroup wrote in this launch.
// PAIRED: this body runs as the SECOND block of a pair launch: it waits for the neighbours' flags before its first x load
// FIRST: this body runs as the FIRST block of a pair launch: its output leaves through write-through stores (sc0 sc1: in memory
// when acknowledged), so that publishing needs no write-back of the whole L2
template <int RT9, int NK1, int DIL, bool PAIRED, bool FIRST = false>
__device__ __forceinline__ void gated_block_fwd2_body(const BlockArgs& a, int ntiles, int tpf, int skip, int* flags, int* timeouts) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64, H = 4 + 7 * DIL, WX = TT + 2 * H, WGW = TT + 8, LDX = 112, LDG = 80, CR = 4 * NK1, LDW = 48;
  constexpr int NCT1 = (WX + 15) / 16;      // column tiles of the h tile (7 at dil 2, 6 at dil 1)
  constexpr int W9P = 728;                  // floats of the packed-tile weights + biases (724, padded to 16 bytes)
  constexpr int NQ = (CR + 7) / 8;          // staged x rows per wave
  static_assert(NCT1 * 16 <= LDX && 79 + 14 * DIL < NCT1 * 16, "h tile must cover every column the k15 taps read");
  float* xs = sm;                            // [CR][LDX]
  float* hs = xs + CR * LDX;                 // [20][LDX]
  float* gs = hs + NARROW * LDX;             // [20][LDG]
  float* w2s = gs + NARROW * LDG;            // [15*20][48]  k15 gate kernels, rows interleaved lin/tanh (see header)
  float* ls = w2s + K15 * NARROW * LDW;      // [20][LDG]  lin   } of this tile, for the row-wise save pass (training forward)
  float* ts = ls + NARROW * LDG;             // [20][LDG]  tanh  }
  float* w9p = ts + NARROW * LDG;            // RT9 = 7: [9][20][4] k9 weights of channels 96..99, then their 4 biases
  const int C = a.C, T = a.T;
  const int Cin = a.Cin;                      // 1 with NK1 == 1: rows 1..3 of the only k-step are zero rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;

  // the first tile's x goes out before the weights: its HBM round trip is the longest latency of the prologue
  // x tile prefetch: wave w owns rows w, w+8, ...; lanes along time (two 64-column halves).  Raw buffer loads: scalar row
  // offset, per-lane time offset, out-of-frame columns pointed past the descriptor so the hardware returns 0 - no
  // address VALU and no value selects (with selects the compiler serialised the loads under register pressure).
  // The x tile travels as float4: lane = (row half lane >> 5, float4 index lane & 31 of the LDX / 4 = 28 per row); a wave
  // instruction covers two rows, wave w owns rows 2w + half + 16 q: 7 x buffer_load_dwordx4 + 7 x ds_write_b128 per lane
  // instead of 26 + 26 dword operations (a VMEM instruction costs the issuing wave ~60 cycles whatever its width).
  constexpr int NQ4 = (CR + 15) / 16;
  f32x4 pf4[NQ4];
  const __amdgpu_buffer_rsrc_t sx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * Cin * T * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, 0, FIRST ? (unsigned)((long)a.B * a.C * T * 4) : 0u, 0x00020000);
  const int pi4 = lane & 31, phalf = lane >> 5;
  int pf_vo = 0;
  auto pf_setup = [&](int tile) {
    const int tl = __builtin_amdgcn_readfirstlane(tile < ntiles ? tile : 0);   // past the end: a harmless re-read
    const int b = tl / tpf, t0 = (tl - b * tpf) * TT;
    const int OOB = 0x7ffffff0;
    // byte offset of (row 2 wave + half, time t0 - H + 4 i4).  A negative time reads the previous row's tail; the staging
    // step masks per element.  Only row 0 of frame 0 can start before the tensor: clamped to 0, shifted when staged.
    pf_vo = pi4 < LDX / 4 ? max(((b * Cin + 2 * wave + phalf) * T + t0 - H + 4 * pi4) * 4, 0) : OOB;
  };
  // (rows >= Cin - the last q of a wave at C = 100 / 50 / 25, all but row 0 of a one-channel input - are not fetched: their
  // lanes' offset goes out of range; the staging step writes zeros there either way)
  auto pf1 = [&](int q) {
    const int vo = (q == NQ4 - 1 && 2 * wave + phalf + 16 * q >= Cin) ? 0x7ffffff0 : pf_vo;
    pf4[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sx, vo, q * 16 * T * 4, PAIRED ? NSC_AUX_COHERENT : NSC_AUX_STREAM));
  };
  auto prefetch = [&](int tile) {
    pf_setup(tile);
#pragma unroll
    for (int q = 0; q < NQ4; ++q) pf1(q);
  };
  // CHAIN (same idea as gated_block_dgrad2_kernel): a workgroup walks CONSECUTIVE tiles [first, last).  From the second tile
  // of a chain on ("steady"), the 14*DIL columns of h and the 8 columns of g that the previous tile already computed are
  // carried over in LDS: phase 1 runs on 4 column tiles instead of 7 (6 at dil 1) and phase 2 on 4 instead of 5.  Every
  // element is produced by the same instruction sequence in both modes, so the output bits do not depend on the mode.
  const int first = (int)((long)blockIdx.x * ntiles / gridDim.x), last = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  NSC_STAMP(32);
  if (!PAIRED) prefetch(first);          // (a paired body: after the neighbours' flags, below - the weights do not wait for them)
  // ---- once per workgroup: weights -> LDS / registers ----
  const int r1 = wave >> 2;
  float w1r[NK1];
  // phase-3 jobs.  C = 50 (RT9 = 4): wave w = row tile w & 3, column tiles 2 (w >> 2) + {0, 1}.  C = 100 (RT9 = 7): SIX dense
  // row tiles - waves 0-3 own row tiles 0-3 (4 column tiles each), waves 4 | 5 share row tile 4 (2 + 2), waves 6 | 7 row tile 5
  // (3 + 1) - and wave 7 also computes channels 96..99 as ONE packed tile [4 time shifts][4 channels] (60 MFMAs; as a seventh
  // padded row tile they cost 180).  Per SIMD (waves w, w + 4): 270 | 270 | 315 | 285 MFMAs; round 2 had 360 | 360 | 360 | 180.
  const int rt3 = RT9 == 7 ? (wave < 4 ? wave : 4 + ((wave - 4) >> 1)) : (wave & 3);
  float w9r[K9][5];
  float b1r[4], b9r[4];
  float b9l;            // bias of output channel rt3 * 16 + l15: the dense phase-3 jobs run transposed (see out_store4)
  // phase-2 jobs: q -> (row tile q % 3, column tile q / 3); wave w runs q = w and w + 8 (wave 7: a discarded duplicate)
  int jrt[2], jct[2];
  bool jlive[2];
  float blr[2][2], brr[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int q = wave + 8 * e;
    jlive[e] = q < 15;
    const int qq = jlive[e] ? q : wave;
    jrt[e] = qq % 3;
    jct[e] = qq / 3;
  }
  if (a.img) {
    // FAST prologue: the engine keeps a kernel-ready IMAGE of the block's parameters (rebuilt once per step by the same
    // gather launch that flips the data-gradient kernels): the LDS image of the k15 gate kernels as it stands, then every
    // lane's register fragments laid out [wave][group of 4][lane][4], so that the whole prologue is ~30 coalesced 16-byte
    // loads per lane instead of ~115 dword loads with index arithmetic (a memory instruction costs the issuing wave 60-100
    // cycles whatever its width: the prologue was 9-13 k cycles of a 40-70 us launch).
    // Round 4: (1) the k15 gate kernels (w2s, 57.6 KB: a quarter of the image) are not read before phase 2 of the first tile -
    // they go straight into LDS by LDS-DMA issued AFTER the wait below (see gated_block_dgrad2_role) and land under phases 0 / 1;
    // (2) fragments that several waves hold in common are ONE copy in the image, fetched by all of them (the second fetch of a
    // line is an L1 hit or merges with the first): group A = W1 fragments + b1 (per row tile of h: 2 copies), group B = k9
    // fragments + b9 (per phase-3 row tile: 6 | 4 copies), group C = the gate biases of a wave's two phase-2 jobs (per wave).
    // 240 -> 170 KB of L2 -> CU traffic per workgroup at C = 100, of which 110 KB are waited for before the first MFMA.
    const f32x4* img4 = reinterpret_cast<const f32x4*>(a.img);
    constexpr int NA4 = K15 * NARROW * LDW / 4, NP4 = RT9 == 7 ? W9P / 4 : 0;
    constexpr int NFA = NK1 + 4, NF4A = (NFA + 3) / 4, NF4B = (K9 * 5 + 4 + 3) / 4, NF4C = 2, NVB = RT9 == 7 ? 6 : 4;
    f32x4 fa[NF4A], fb[NF4B], fc[NF4C];
    const f32x4* fbase = img4 + NA4 + NP4 + lane;
#pragma unroll
    for (int g = 0; g < NF4A; ++g) fa[g] = fbase[(r1 * NF4A + g) * 64];
#pragma unroll
    for (int g = 0; g < NF4B; ++g) fb[g] = fbase[(2 * NF4A + rt3 * NF4B + g) * 64];
#pragma unroll
    for (int g = 0; g < NF4C; ++g) fc[g] = fbase[(2 * NF4A + NVB * NF4B + wave * NF4C + g) * 64];
    f32x4 tP = {0.f, 0.f, 0.f, 0.f};
    if (RT9 == 7 && tid < W9P / 4) tP = img4[NA4 + tid];
    if (RT9 == 7 && tid < W9P / 4) reinterpret_cast<f32x4*>(w9p)[tid] = tP;
#pragma unroll
    for (int u = 0; u < NK1; ++u) w1r[u] = fa[u / 4][u % 4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) b1r[reg] = fa[(NK1 + reg) / 4][(NK1 + reg) % 4];
#pragma unroll
    for (int tap = 0; tap < K9; ++tap)
#pragma unroll
      for (int u = 0; u < 5; ++u) w9r[tap][u] = fb[(tap * 5 + u) / 4][(tap * 5 + u) % 4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) b9r[reg] = fb[(45 + reg) / 4][(45 + reg) % 4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        blr[e][u] = fc[0][2 * e + u];
        brr[e][u] = fc[1][2 * e + u];
      }
  } else {
  // Every load below uses a CLAMPED index instead of a mask: pad rows of A (output channels >= 20 / >= C) only feed
  // output rows that are never stored, and pad k-rows (ci >= C) multiply x rows that phase 0 writes as zeros, so any
  // finite stand-in value is harmless - and without selects all ~125 loads per lane are in flight at once.
  {
    constexpr int NE = (K15 * NARROW * LDW + 511) / 512;
    float tmp[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = min(tid + 512 * i, K15 * NARROW * LDW - 1);
      const int row = e / LDW, r = e - row * LDW;
      const int ii = r & 15;
      const int c = min((r >> 4) * 8 + (ii >> 2) * 2 + (ii & 1), NARROW - 1);
      const float* src = (ii & 2) ? a.wr : a.wl;
      tmp[i] = src[row * NARROW + c];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 512 * i;
      if (e < K15 * NARROW * LDW) w2s[e] = tmp[i];
    }
  }
  // phase 1: a wave owns ONE of the two row tiles of h (waves 0-3: channels 0..15, waves 4-7: 16..19 + padding), so it
  // keeps NK1 fragments of W1, not 2 NK1 (the second set cost 25 registers and pushed the C = 100 kernel into scratch)
#pragma unroll
  for (int u = 0; u < NK1; ++u) w1r[u] = a.w1[min(4 * u + kq, Cin - 1) * NARROW + min(r1 * 16 + l15, NARROW - 1)];
#pragma unroll
  for (int tap = 0; tap < K9; ++tap)
#pragma unroll
    for (int u = 0; u < 5; ++u) w9r[tap][u] = a.w9[(tap * NARROW + 4 * u + kq) * C + min(rt3 * 16 + l15, C - 1)];
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    b1r[reg] = a.b1[min(r1 * 16 + kq * 4 + reg, NARROW - 1)];
    b9r[reg] = a.b9[min(rt3 * 16 + kq * 4 + reg, C - 1)];
  }
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = jrt[e] * 8 + kq * 2 + u;
      blr[e][u] = a.bl[min(c, NARROW - 1)];
      brr[e][u] = a.br[min(c, NARROW - 1)];
    }
  if (RT9 == 7) {
    for (int e = tid; e < W9P; e += 512)
      w9p[e] = e < K9 * NARROW * 4 ? a.w9[(e >> 2) * C + 96 + (e & 3)] : (e < K9 * NARROW * 4 + 4 ? a.b9[96 + e - K9 * NARROW * 4] : 0.f);
  }
  }


  if (PAIRED) {
    nsc_pair_wait(flags, timeouts);                // the first block's output around this workgroup's tiles is in memory
    prefetch(first);
  }
  nsc_wait_vmem();   // weights (and the first tile) are in: no vmcnt guards on register operands inside the loop
  {
    // b9r[reg] = b9[rt3 * 16 + 4 kq + reg] (the fragment layout of the image); the transposed phase 3 wants b9[rt3 * 16 + l15]:
    // lane (kq' = l15 >> 2, any l15') holds it in register l15 & 3
    float tb[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) tb[reg] = __shfl(b9r[reg], (l15 >> 2) * 16, 64);
    b9l = (l15 & 2) ? ((l15 & 1) ? tb[3] : tb[2]) : ((l15 & 1) ? tb[1] : tb[0]);
  }
  // the k15 gate kernels by LDS-DMA (image path; see gated_block_dgrad2_role): the youngest vector-memory operations when the tile
  // loop starts, waited for by hand before the first tile's phase 2
  const bool dma2 = a.img != nullptr;
  if (dma2) {
    constexpr int NA4 = K15 * NARROW * LDW / 4, NG2 = (NA4 + 511) / 512;
    const unsigned lds2 = (unsigned)(unsigned long long)w2s;
#pragma unroll
    for (int i = 0; i < NG2; ++i) {
      if (tid + 512 * i < NA4) {
        const f32x4* gp = reinterpret_cast<const f32x4*>(a.img) + tid + 512 * i;
        const unsigned m0v = lds2 + (unsigned)((i * 8 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m0v) : "memory", "m0");
      }
    }
  }
  NSC_STAMP(33);
  for (int tile = first; tile < last; ++tile) {
    const int b = tile / tpf, t0 = (tile - b * tpf) * TT;
    const bool fresh = tile == first || t0 == 0;                                   // workgroup-uniform
    const bool next_steady = tile + 1 < last && (tile + 1) - ((tile + 1) / tpf) * tpf != 0;
    NSC_STAMP(34);
    if (!fresh) {
      // carried g: columns [64, 72) of the previous tile are columns [0, 8) of this one (phase 3 of the previous tile is
      // behind the loop-end barrier; phase 2 of this tile writes columns >= 8 only)
      if (tid < NARROW * 8) {
        const int r = tid >> 3, cidx = tid & 7;
        gs[r * LDG + cidx] = gs[r * LDG + cidx + TT];
      }
      // carried h: the k15 convs of this tile read h from column 8 on; its columns [8, 2H) are the previous tile's
      // [72, 2H + 64).  (Here, not at the end of the previous tile: its save pass reads those columns during phase 3.)
      for (int e = tid; e < NARROW * 14 * DIL; e += 512) {
        const int r = e / (14 * DIL), cidx = 8 + (e - r * (14 * DIL));
        hs[r * LDX + cidx] = hs[r * LDX + cidx + TT];
      }
    }
    // ---- phase 0: prefetched x tile -> LDS (out-of-frame elements are the conv's zero padding) ----
    if (pi4 < LDX / 4) {
      const int tb = t0 - H + 4 * pi4;
      const bool m0 = (unsigned)tb < (unsigned)T, m1 = (unsigned)(tb + 1) < (unsigned)T;
      const bool m2 = (unsigned)(tb + 2) < (unsigned)T, m3 = (unsigned)(tb + 3) < (unsigned)T;
      const bool clamped = b == 0 && wave == 0 && phalf == 0 && tb < 0 && tb > -4;
#pragma unroll
      for (int q = 0; q < NQ4; ++q) {
        const int r = 2 * wave + phalf + 16 * q;
        if (r < CR) {
          f32x4 v = pf4[q];
          if (clamped) {
            // these lanes' offset for the very first row of the tensor was negative and was clamped to time 0 (the clamp
            // holds for all their rows: the row enters through the scalar offset): element e holds time e, wanted tb + e
            const f32x4 w = v;
            const int sh = -tb;
            v[1] = sh == 1 ? w[0] : 0.f;
            v[2] = sh == 1 ? w[1] : (sh == 2 ? w[0] : 0.f);
            v[3] = sh == 1 ? w[2] : (sh == 2 ? w[1] : w[0]);
          }
          const bool live = r < Cin;                         // pad rows of the last k-step are zeros
          v[0] = live && m0 ? v[0] : 0.f;
          v[1] = live && m1 ? v[1] : 0.f;
          v[2] = live && m2 ? v[2] : 0.f;
          v[3] = live && m3 ? v[3] : 0.f;
          *reinterpret_cast<f32x4*>(xs + r * LDX + 4 * pi4) = v;
        }
      }
    }
    NSC_STAMP(35);
    nsc_lds_barrier();
    NSC_STAMP(36);
    // The next tile's x: one load at a time from inside the k15 loop (issued in one piece here, the NQ4 loads of all 8 waves
    // of every workgroup cost ~1 k cycles of phase 1 with the matrix pipe idle: gated_block_dgrad2_kernel's pf_setup).  Probe runs that skip
    // phases issue them in one piece.
    const bool spread = skip == 0;
    pf_setup(tile + 1 < last ? tile + 1 : tile);
    if (!spread && !(skip & 8)) {
#pragma unroll
      for (int q = 0; q < NQ4; ++q) pf1(q);
    }

    // ---- phase 1: h = lrelu(W1 x + b1); column tile = wave ----
    if (!(skip & 1)) {
      // column tiles of this wave: steady - the TT new columns [2H, 2H + TT), one tile per wave; fresh - all NCT1 tiles from
      // column 0, two per wave (the 8th job of a row tile duplicates the 7th: same values, harmless)
      const int jb0 = fresh ? (wave & 3) * 16 : 2 * H + (wave & 3) * 16;
      const int jb1 = min((wave & 3) + 4, NCT1 - 1) * 16;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* xc0 = xs + kq * LDX + jb0 + l15;
      const float* xc1 = xs + kq * LDX + jb1 + l15;
      if (fresh) {
#pragma unroll
        for (int u = 0; u < NK1; ++u) {
          acc0 = mfma4(w1r[u], xc0[4 * u * LDX], acc0);
          acc1 = mfma4(w1r[u], xc1[4 * u * LDX], acc1);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NK1; ++u) acc0 = mfma4(w1r[u], xc0[4 * u * LDX], acc0);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && !fresh) break;
        const int j = (e ? jb1 : jb0) + l15;
        const int t = t0 - H + j;
        const bool live = j < WX && t >= 0 && t < T;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int o = r1 * 16 + kq * 4 + reg;
          if (o < NARROW) {
            float v = (e ? acc1[reg] : acc0[reg]) + b1r[reg];
            v = v > 0.f ? v : NSC_LRELU_ALPHA * v;
            hs[o * LDX + j] = live ? v : 0.f;
          }
        }
      }
    }
    NSC_STAMP(37);
    // (first tile on the image path: the LDS-DMA of the k15 gate kernels has had phases 0 and 1 to land; nothing younger is
    // outstanding in the spread form - the next tile's x loads go out inside phase 2)
    if (dma2 && tile == first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    nsc_lds_barrier();
    NSC_STAMP(38);

    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));               // (addresses of the save pass are recomputed per tile, not held in registers)
    const bool tvec = (T & 3) == 0;
    auto store4 = [&](float* gp, const f32x4& v, int t) {
      if (tvec) {
        if (t < T) nsc_store4_late(gp, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e < T) gp[e] = v[e];
      }
    };
    auto save_h = [&](int it0, int step) {
#pragma unroll 2
      for (int it = it0; it < NARROW * 16; it += step) {
        const int c = it >> 4, t = t0 + 4 * (it & 15);
        const float* hp = hs + c * LDX + H + 4 * (it & 15);
        f32x4 v;
        if (H % 2 == 0) {
          const float2 v0 = *reinterpret_cast<const float2*>(hp), v1 = *reinterpret_cast<const float2*>(hp + 2);
          v = (f32x4){v0.x, v0.y, v1.x, v1.y};
        } else {
          v = (f32x4){hp[0], hp[1], hp[2], hp[3]};
        }
        store4(a.h_out + ((long)b * NARROW + c) * T + t, v, t);
      }
    };
    auto save_lg = [&](int it0, int step) {
      const int klo = fresh ? 1 : 2, khi = next_steady ? WGW / 4 : 1 + TT / 4;      // float4 groups of columns [4 klo, 4 khi)
#pragma unroll 2
      for (int it = it0; it < NARROW * 18; it += step) {
        const int c = it / 18, k = it - 18 * c;
        const int t = t0 - 4 + 4 * k;
        if (k >= klo && k < khi) {
          const long gi = ((long)b * NARROW + c) * T + t;
          store4(a.lin_out + gi, *reinterpret_cast<const f32x4*>(ls + c * LDG + 4 * k), t);
          store4(a.th_out + gi, *reinterpret_cast<const f32x4*>(ts + c * LDG + 4 * k), t);
          store4(a.g_out + gi, *reinterpret_cast<const f32x4*>(gs + c * LDG + 4 * k), t);
        }
      }
    };
    // ---- phase 2: both k15 gate convs, A from LDS (w2s), B from LDS (hs) ----
    if (!(skip & 2)) {
      // fresh: 15 jobs (3 row tiles x 5 column tiles from column 0); steady: 12 jobs (4 column tiles from column 8: the
      // first 8 columns of g are carried) - waves 4-7 then run ONE job
      const int joff = fresh ? 0 : 8;
      const bool two = fresh || wave < 4;                 // wave-uniform: does this wave run a second job?
      f32x4 acc[2];
      acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* ab0 = w2s + kq * LDW + jrt[0] * 16 + l15;
      const float* ab1 = w2s + kq * LDW + jrt[1] * 16 + l15;
      const float* hb0 = hs + kq * LDX + jct[0] * 16 + l15 + joff;
      const float* hb1 = hs + kq * LDX + jct[1] * 16 + l15 + joff;
      // load q of the next tile's x goes out before tap 2 q + 1
      auto hook = [&](int tap) {
        if ((tap & 1) && tap / 2 < NQ4 && spread) {
          __builtin_amdgcn_sched_barrier(0);
          pf1(tap / 2);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      static_assert(2 * NQ4 - 1 < K15, "one x load per odd tap");
      if (two) {
#pragma unroll
        for (int tap = 0; tap < K15; ++tap) {
          hook(tap);
#pragma unroll
          for (int u = 0; u < 5; ++u) {
            const int ao = (tap * NARROW + 4 * u) * LDW, ho = 4 * u * LDX + tap * DIL;
            acc[0] = mfma4(ab0[ao], hb0[ho], acc[0]);
            acc[1] = mfma4(ab1[ao], hb1[ho], acc[1]);
          }
        }
      } else {
#pragma unroll
        for (int tap = 0; tap < K15; ++tap) {
          hook(tap);
#pragma unroll
          for (int u = 0; u < 5; ++u) {
            const int ao = (tap * NARROW + 4 * u) * LDW, ho = 4 * u * LDX + tap * DIL;
            acc[0] = mfma4(ab0[ao], hb0[ho], acc[0]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!jlive[e] || (e == 1 && !two)) continue;
        const int jj = jct[e] * 16 + l15 + joff;
        const int c0 = jrt[e] * 8 + kq * 2;
        const int t = t0 - 4 + jj;
        const bool live = jj < WGW && t >= 0 && t < T;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = c0 + u;
          if (c < NARROW) {
            const float lin = acc[e][u] + blr[e][u];
            const float th = nsc_tanh(acc[e][2 + u] + brr[e][u]);
            gs[c * LDG + jj] = live ? lin * th : 0.f;
            if (a.lin_out) {                  // kept for the backward pass: leave row-wise in the save pass below
              ls[c * LDG + jj] = lin;
              ts[c * LDG + jj] = th;
            }
          }
        }
      }
    }
    NSC_STAMP(39);
    nsc_lds_barrier();
    NSC_STAMP(40);
    // ---- save pass (training forward): h, lin, tanh, g leave for HBM row-wise, 16 bytes per lane, whole 256-B lines, from
    // LDS - not from the accumulator layout (4-byte stores in 64-byte pieces, ~12 per lane and phase).  h: this tile's own 64
    // steps (columns [H, H + 64): the left part was carried from the previous tile); lin / tanh / g: the columns phase 2
    // produced that the next tile will not produce again.  Store issue is what costs (a wave's VMEM instruction takes 100+
    // cycles to issue under load), so the pass runs where the matrix pipe does not wait for it, in phase 3: at C = 100 waves
    // 4 and 5 - who have half a row tile of MFMAs there, their SIMD partners a whole one - do half of it each, before their
    // MFMAs; at C = 50 every wave does its share, waves 0-3 before their MFMAs and waves 4-7 (their SIMD partners) after.
    // (h from waves 4-7 at the end of phase 2, where they run one k15 job against the two of waves 0-3, was slower: that
    // "slack" is their SIMD partners' MFMA time.  One wave doing the whole pass takes ~10 k cycles.)
    auto save_pass = [&](int it0, int step) {
      if (a.h_out) save_h(it0, step);
      if (a.lin_out) save_lg(it0, step);
    };
    if (RT9 == 7) {
      if (wave == 4 || wave == 5) save_pass(tid_s - 256, 128);
    } else if (wave < 4) {
      save_pass(tid_s, 512);
    }

    // ---- phase 3: y = W9 * g + b9 + x; this wave's row tile (weights in registers), NC3 column tiles ----
    // (lane-derived indices of this phase from an opaque copy: the compiler otherwise hoists each role's addresses out of the
    // tile loop and holds them in registers across all phases)
    int lane3 = lane;
    asm volatile("" : "+v"(lane3));
    const int l15p = lane3 & 15, kqp = lane3 >> 4;
    // (first block of a pair launch: the output is written through to memory, see nsc_pair_publish)
    auto out_store = [&](float v, int b_, int o, int t) {
      if constexpr (FIRST) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), sout, ((b_ * C + o) * T + t) * 4, 0, NSC_AUX_COHERENT);
      } else {
        a.out[((long)b_ * C + o) * T + t] = v;
      }
    };
    // The dense jobs compute the TRANSPOSED product (the two MFMA operands swapped: rows of D = time, columns = output channels;
    // both operands keep their lane layout, 16x16x4 fragments of A and B are mirror images): a lane's four accumulator registers
    // are four CONSECUTIVE steps of one channel and leave as one 16-byte store - 4 store instructions per lane and job instead of 16.
    auto out_store4 = [&](const f32x4& v, int b_, int o, int t) {
      float* gp = a.out + ((long)b_ * C + o) * T + t;
      if constexpr (FIRST) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), sout, ((b_ * C + o) * T + t) * 4, 0, NSC_AUX_COHERENT);   // (pairs: T % 4 == 0)
      } else if (tvec && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {      // (rows 16-byte aligned: T % 4 == 0 and an aligned tensor)
        *reinterpret_cast<f32x4*>(gp) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e < T) gp[e] = v[e];
      }
    };
    // dense job: this wave's row tile, NC column tiles from ct0 on (weights in registers)
    auto dense3 = [&](auto nc_c, int ct0) {
      constexpr int NC = decltype(nc_c)::value;
      f32x4 acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* gb = gs + kqp * LDG + ct0 * 16 + l15p;
#pragma unroll
      for (int tap = 0; tap < K9; ++tap) {
#pragma unroll
        for (int u = 0; u < 5; ++u)
#pragma unroll
          for (int c = 0; c < NC; ++c) acc[c] = mfma4(gb[4 * u * LDG + c * 16 + tap], w9r[tap][u], acc[c]);   // (transposed product)
      }
      const int o = rt3 * 16 + l15p;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int tt = (ct0 + c) * 16 + 4 * kqp;
        const int t = t0 + tt;
#if defined(NSC_EXP) && (NSC_EXP & 16)
        if (o < C && t < T && a.B < 0) {                   // timing experiment: the output epilogue never stores
#else
        if (o < C && t < T) {
#endif
          const float* xr = xs + (NK1 == 1 ? 0 : o) * LDX + H + tt;             // Cin = 1: broadcast residual
          f32x4 v;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            float u_ = acc[c][reg] + b9l + xr[reg];
            if (!a.flat) u_ = u_ > 0.f ? u_ : NSC_LRELU_ALPHA * u_;
            v[reg] = u_;
          }
          out_store4(v, b, o, t);
        }
      }
    };
    // channels 96..99 of a C = 100 block, PACKED as in gated_block_dgrad2_kernel's d9_packed: row (s, i), column n ->
    // y[96 + i][4n + s] = sum_{m, c} A[(s,i)][(m,c)] g[c][4n + m], A = w9[m - s][c][96 + i] for 0 <= m - s < 9, else 0
    auto packed3 = [&]() {
      const int sft = l15p >> 2, ich = l15p & 3;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* gp = gs + kqp * LDG + 4 * l15p;
#pragma unroll
      for (int m = 0; m < K9 + 3; ++m) {
        const int tap = m - sft;
        const bool ok = (m >= 3 && m <= K9 - 1) || (unsigned)tap < (unsigned)K9;
        const float* ap = w9p + (ok ? tap : 0) * (NARROW * 4) + kqp * 4 + ich;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
          const float av = ap[16 * u];
          const float a_ = ok ? av : 0.f;
          if ((m * 5 + u) & 1) acc1 = mfma4(a_, gp[4 * u * LDG + m], acc1);
          else acc0 = mfma4(a_, gp[4 * u * LDG + m], acc0);
        }
      }
      const int tt = 4 * l15p + kqp, t = t0 + tt;
      if (t < T) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int o = 96 + reg;
          float v = acc0[reg] + acc1[reg] + w9p[K9 * NARROW * 4 + reg] + xs[(NK1 == 1 ? 0 : o) * LDX + H + tt];
          if (!a.flat) v = v > 0.f ? v : NSC_LRELU_ALPHA * v;
          out_store(v, b, o, t);
        }
      }
    };
    if (!(skip & 4)) {
      if (RT9 == 7) {
        if (wave < 4) dense3(std::integral_constant<int, 4>{}, 0);
        else if (wave < 6) dense3(std::integral_constant<int, 2>{}, 2 * (wave - 4));
        else if (wave == 6) dense3(std::integral_constant<int, 3>{}, 0);
        else {
          dense3(std::integral_constant<int, 1>{}, 3);
          packed3();
        }
      } else {
        dense3(std::integral_constant<int, 2>{}, 2 * (wave >> 2));
      }
    }
    if (RT9 != 7 && wave >= 4) save_pass(tid_s, 512);
    NSC_STAMP(41);
    nsc_lds_barrier();   // xs / hs / gs are rewritten by the next tile
    NSC_STAMP(42);
  }
  NSC_STAMP(43);
}

template <int RT9, int NK1, int DIL>
__global__ __launch_bounds__(512) void gated_block_fwd2_kernel(BlockArgs a, int ntiles, int tpf, int skip) {
  gated_block_fwd2_body<RT9, NK1, DIL, false>(a, ntiles, tpf, skip, nullptr, nullptr);
}

// Two consecutive blocks of a stack (dilation 1 then 2, the reference's `_stack_bottleneck_blocks`: neural_speech_coding_module.py
// :183-217) in ONE launch: see nsc_pair_publish / nsc_pair_wait.  a1.x must be a0.out.
// (NK1A = 1: the first block has ONE input channel - the first stack of a decoder, whose input is the quantised code)
template <int RT9, int NK1A, int NK1B>
__global__ __launch_bounds__(512) void gated_block_fwd2_pair_kernel(BlockArgs a0, BlockArgs a1, int ntiles, int tpf, int* flags, int* timeouts) {
  gated_block_fwd2_body<RT9, NK1A, 1, false, true>(a0, ntiles, tpf, 0, nullptr, nullptr);
  nsc_pair_publish(flags);
  gated_block_fwd2_body<RT9, NK1B, 2, true>(a1, ntiles, tpf, 0, flags, timeouts);
}

template <int RT9, int NK1, int DIL>
static int launch_block_fwd2(const BlockArgs& a, hipStream_t st) {
  constexpr int CR = 4 * NK1;
  const size_t smem = ((size_t)(CR + NARROW) * 112 + (size_t)3 * NARROW * 80 + (size_t)K15 * NARROW * 48 + (RT9 == 7 ? 728 : 0)) * sizeof(float);
  auto kern = gated_block_fwd2_kernel<RT9, NK1, DIL>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);   // once per instantiation
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd2: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, 64);
  const int ntiles = a.B * tpf;
  static const int skip = NSC_PROBE_INT("NSC_FWD2_SKIP", 0);   // timing probe only
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, 256)), dim3(512), smem, st, a, ntiles, tpf, skip);
  NSC_CHECK_LAUNCH("gated_block_fwd2");
  return NSC_OK;
}

extern "C" int nsc_gated_block_fwd(const float* x, const float* w1, const float* b1, const float* wl, const float* bl,
                                   const float* wr, const float* br, const float* w9, const float* b9, float* out,
                                   float* h_out, float* lin_out, float* th_out, float* g_out, int B, int C, int T,
                                   int narrow, int k9, int dil, int flat, void* stream) {
  NSC_REQUIRE(x && w1 && b1 && wl && bl && wr && br && w9 && b9 && out, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd: null pointer");
  NSC_REQUIRE(B > 0 && C > 1 && T > 0 && dil > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd: bad sizes");
  NSC_REQUIRE(narrow == NARROW && k9 == K9, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_fwd: built for narrow=20, k9=9 (got %d, %d): use the unfused path", narrow, k9);
  NSC_REQUIRE(C <= 112 && dil <= 4, NSC_ERR_UNSUPPORTED, "nsc_gated_block_fwd: C %d > 112 or dil %d > 4", C, dil);
  NSC_REQUIRE(!(lin_out || th_out || g_out) || (lin_out && th_out && g_out), NSC_ERR_BAD_ARG,
              "nsc_gated_block_fwd: lin/th/g outputs must be given together");
  const int H = 4 + 7 * dil, WX = 64 + 2 * H;
  int ldx = WX;
  while ((ldx & 31) != 16) ++ldx;
  if (ldx < 80) ldx = 80;
  int ldg = 80;  // 72 columns used, 80 % 32 == 16
  const int C4 = (C + 3) & ~3;
  const size_t smem = ((size_t)(C4 + NARROW) * ldx + (size_t)NARROW * ldg) * sizeof(float);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "nsc_gated_block_fwd: %zu B LDS", smem);
  BlockArgs a{B, C, T, dil, flat, x, w1, b1, wl, bl, wr, br, w9, b9, out, h_out, lin_out, th_out, g_out, C, nullptr};
  dim3 grid(nsc_cdiv(T, 64), B);
  hipStream_t st = (hipStream_t)stream;
  static const bool v1_only = NSC_PROBE_SET("NSC_BLOCK_FWD_V1");   // A/B switch for profiling
  if (!v1_only && (dil == 1 || dil == 2)) {
    if (C == 100) return dil == 1 ? launch_block_fwd2<7, 25, 1>(a, st) : launch_block_fwd2<7, 25, 2>(a, st);
    if (C == 50) return dil == 1 ? launch_block_fwd2<4, 13, 1>(a, st) : launch_block_fwd2<4, 13, 2>(a, st);
    // C = 25 (the third resolution of '2 2' codecs) on the C = 50 job table: two of its four phase-3 row tiles are padding,
    // but the weights are stationary and the tiles chained - the per-tile kernel fetched every A fragment from global memory
    if (C == 25) return dil == 1 ? launch_block_fwd2<4, 7, 1>(a, st) : launch_block_fwd2<4, 7, 2>(a, st);
  }
  const int nrt = nsc_cdiv(C, 16);
#define LAUNCH_BLK(RT)                                                                                              \
  do {                                                                                                              \
    auto kern = gated_block_fwd_kernel<RT>;                                                                         \
    if (smem > 64 * 1024) {                                                                                         \
      const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024); \
      NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd: smem attr: %s", hipGetErrorString(e));         \
    }                                                                                                               \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a, ldx, ldg);                                               \
  } while (0)
  if (nrt <= 4) LAUNCH_BLK(4);
  else LAUNCH_BLK(7);
#undef LAUNCH_BLK
  NSC_CHECK_LAUNCH("gated_block_fwd");
  return NSC_OK;
}


// The first block of each decoder stage has ONE input channel (the quantised code): its 1x1 conv is 1 -> 20 and the
// residual add broadcasts x [B,1,T] over the `C` output channels (nn_core_operator.py:110-112 with a [B,T,1] input).
// Same persistent kernel with a single k-step in phase 1 (rows 1..3 of the staged x tile are zero rows).
extern "C" int nsc_gated_block_fwd_cin1(const float* x, const float* w1, const float* b1, const float* wl, const float* bl,
                                        const float* wr, const float* br, const float* w9, const float* b9, float* out,
                                        float* h_out, float* lin_out, float* th_out, float* g_out, int B, int C, int T,
                                        int narrow, int k9, int dil, int flat, void* stream) {
  NSC_REQUIRE(x && w1 && b1 && wl && bl && wr && br && w9 && b9 && out, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_cin1: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_cin1: bad sizes");
  NSC_REQUIRE(narrow == NARROW && k9 == K9 && (C == 100 || C == 50 || C == 25) && (dil == 1 || dil == 2), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_fwd_cin1: built for narrow=20, k9=9, C in {100, 50, 25}, dil in {1, 2} (got %d, %d, %d, %d)", narrow, k9, C, dil);
  NSC_REQUIRE(!(lin_out || th_out || g_out) || (lin_out && th_out && g_out), NSC_ERR_BAD_ARG,
              "nsc_gated_block_fwd_cin1: lin/th/g outputs must be given together");
  BlockArgs a{B, C, T, dil, flat, x, w1, b1, wl, bl, wr, br, w9, b9, out, h_out, lin_out, th_out, g_out, 1, nullptr};
  hipStream_t st = (hipStream_t)stream;
  if (C == 100) return dil == 1 ? launch_block_fwd2<7, 1, 1>(a, st) : launch_block_fwd2<7, 1, 2>(a, st);
  return dil == 1 ? launch_block_fwd2<4, 1, 1>(a, st) : launch_block_fwd2<4, 1, 2>(a, st);
}

// =====================================================================================================
// Persistent weight-gradient kernel of one gated block: all eight parameter gradients from the SAVED
// activations (x, h, g) and the data-path gradients (dy, dlin, dgate, dz1), which the per-conv backward has
// already produced.  One workgroup per CU walks (frame, 64-step tile) pairs, both MFMA operands come from LDS
// (k = time), the 160 accumulator registers per lane persist across all tiles and are flushed once.  The flush
// order is rotated per workgroup so concurrent float atomics hit different addresses.
//   dW9[tap,ci,o] += g[ci,t+tap-4] dy[o,t]      dWl/dWr[tap,ci,c] += h[ci,t+(tap-7)d] dlin/dgate[c,t]
//   dW1[ci,o]     += x[ci,t] dz1[o,t]           bias gradients via a row of ones
// =====================================================================================================
#ifndef NSC_WG_UNROLL
#define NSC_WG_UNROLL 4   // k-steps of the MFMA loop unrolled together: at 1 every step exposed an LDS round trip (-8 % launch time at 4; 8, 16: same)
#endif

__device__ __forceinline__ void wg_flush(float* p, float v, bool plain) {
  if (plain) *p = v;
  else atomicAdd(p, v);
}

// NW = waves per workgroup.  8: two waves per SIMD (fastest alone, but it fills the register file so nothing can share
// the CU).  4: one wave per SIMD with <= 256 registers, leaving half of the registers and 76 KB of LDS per CU to the
// data-gradient kernels that run concurrently on the main stream.
// PART selects which gradients the launch produces: 0 = all eight (+ optional 1x1 data gradient); 1 = dW9/db9 only
// (stages g and dy: 32 KB, ~90 registers); 2 = dWl/dWr/dW1 + biases (stages h, da, x, dz1: 50 KB).  The two light
// parts leave most of a CU's registers and LDS to the data-gradient kernels running concurrently on the main stream.
// wg / nwg: this workgroup's index among the nwg workgroups that share the block's tiles; slab_id: its slab in the workspace
template <int RT9, int NW, int PART>
__device__ __forceinline__ void block_wgrad_body(const BlockWgradArgs& a, int ldn, int ldg, int ldh, int wg, int nwg,
                                                 int slab_id) {
  constexpr bool P9 = PART != 2, PLR = PART != 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64;
  const int C = a.C, T = a.T, d = a.dil;
  const int Cx = a.Cin;                        // rows of x (the fused D1 below is for Cx == C only)
  const int Hh = 7 * d;
  // rows Cx / Cx+1 of xn are the zero / ones rows used by every part (part 1 keeps only those two rows of xn)
  float* xn = sm;                              // [Cx + 2][ldn]     rows Cx = zeros, Cx+1 = ones
  float* dys = xn + (Cx + 2) * ldn;            // [C][ldn]                              (parts 0, 1)
  float* gs = dys + (P9 ? C * ldn : 0);        // [NARROW + 2][ldg] g on [t0-4, t0+68)  (parts 0, 1)
  float* hs = gs + (P9 ? (NARROW + 2) * ldg : 0);   // [NARROW + 2][ldh] h on [t0-Hh, t0+64+Hh)   (parts 0, 2)
  float* dl = hs + (PLR ? (NARROW + 2) * ldh : 0);  // [NARROW][ldn] dlin  (dgate follows: dg_ = dl + NARROW*ldn)
  float* dg_ = dl + NARROW * ldn;              // [NARROW][ldn] dgate
  float* dhs = dg_ + NARROW * ldn;             // [NARROW][ldn] dz1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;

  // 8 waves (two per SIMD): each owns fewer accumulator tiles (100 registers) and the partner wave hides LDS /
  // global latency that a single wave per SIMD exposed.
  for (int j = tid; j < ldn; j += 64 * NW) { xn[Cx * ldn + j] = 0.f; xn[(Cx + 1) * ldn + j] = 1.f; }
  if (PLR) for (int j = tid; j < ldh; j += 64 * NW) { hs[NARROW * ldh + j] = 0.f; hs[(NARROW + 1) * ldh + j] = 1.f; }
  if (P9) for (int j = tid; j < ldg; j += 64 * NW) { gs[NARROW * ldg + j] = 0.f; gs[(NARROW + 1) * ldg + j] = 1.f; }

  constexpr int R9 = P9 ? (12 + NW - 1) / NW : 1, RLR = PLR ? (19 + NW - 1) / NW : 1,
                R1 = PLR ? (RT9 + NW - 1) / NW : 1;   // row tiles per wave (1 = dummy for a disabled part)
  f32x4 g9[R9][RT9], glr[RLR][3], g1[R1][2];   // row tiles {w, w+NW, ...}
#pragma unroll
  for (int r = 0; r < R9; ++r)
#pragma unroll
    for (int c = 0; c < RT9; ++c) g9[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RLR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) glr[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R1; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) g1[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int off9[R9], offlr[RLR], off1[R1];
#pragma unroll
  for (int r = 0; r < R9; ++r) {
    const int kk = (wave + NW * r) * 16 + l15;
    if (kk < K9 * NARROW) { const int tap = kk / NARROW, ci = kk - tap * NARROW; off9[r] = ci * ldg + tap; }
    else off9[r] = (kk == K9 * NARROW ? (NARROW + 1) : NARROW) * ldg;
  }
#pragma unroll
  for (int r = 0; r < RLR; ++r) {
    const int kk = (wave + NW * r) * 16 + l15;
    if (kk < K15 * NARROW) { const int tap = kk / NARROW, ci = kk - tap * NARROW; offlr[r] = ci * ldh + tap * d; }
    else offlr[r] = (kk == K15 * NARROW ? (NARROW + 1) : NARROW) * ldh;
  }
#pragma unroll
  for (int r = 0; r < R1; ++r) {
    const int ci = (wave + NW * r) * 16 + l15;
    off1[r] = (ci < Cx ? ci : (ci == Cx ? Cx + 1 : Cx)) * ldn;
  }
  // B-operand offsets are relative to `sm`; columns that do not exist point at the zero row of xn (no masks needed)
  const int zrow = (int)(xn - sm) + Cx * ldn;
  int offb_lr[3], offb9[RT9];
#pragma unroll
  for (int ct = 0; ct < 3; ++ct) {
    const int c = ct * 8 + (l15 >> 2) * 2 + (l15 & 1);
    offb_lr[ct] = c < NARROW ? (int)(dl - sm) + ((l15 & 2) ? NARROW * ldn : 0) + c * ldn : zrow;
  }
#pragma unroll
  for (int c = 0; c < RT9; ++c) {
    const int o = c * 16 + l15;
    offb9[c] = o < C ? (int)(dys - sm) + o * ldn : zrow;
  }
  const int offb1 = (16 + l15 < NARROW) ? (int)(dhs - sm) + (16 + l15) * ldn : zrow;

  // ---- software-pipelined staging: the NEXT tile's global loads are issued into registers before the MFMA loop of the
  // current tile and written to LDS after it (one wave per SIMD cannot hide load latency any other way).  Row r of a
  // region is owned by wave r%4; lanes walk time.  All loads are unconditional (clamped) + value select.
  constexpr int QX = RT9 * 16 / NW;                 // rows per wave for the C-channel tensors (x, dy)
  constexpr int QA = 40 / NW, QN = (NARROW + NW - 1) / NW;
  float rx[QX], ry[QX], ra[QA], rz[QN], rg[QN][2], rh[QN][2];
  // Raw buffer loads: the descriptors are wave-uniform, the row offset is a SCALAR (soffset) and the per-lane offset is
  // just the time index, so the ~50 prefetch loads need no per-load address registers; out-of-frame time indices are
  // sent out of the descriptor's range and come back as 0 from the hardware bounds check (no value selects either).
  const unsigned nbC = (unsigned)((long)a.B * C * T * 4), nbN = (unsigned)((long)a.B * NARROW * T * 4);
  const __amdgpu_buffer_rsrc_t sx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * Cx * T * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, nbC, 0x00020000);
  const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.da), 0, 2 * nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz1), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h), 0, nbN, 0x00020000);
  auto bl = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  auto load_tile = [&](int tile) {
    const int b = __builtin_amdgcn_readfirstlane(tile / a.tiles_per_frame);
    const int t0 = __builtin_amdgcn_readfirstlane((tile - b * a.tiles_per_frame) * TT);
    const int OOB = 0x7ffffff0;                       // byte offset past every descriptor -> load returns 0
    const int t = t0 + lane;
    const int vt = t < T ? t * 4 : OOB;
    int vg[2], vh[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = lane + 64 * jj;
      const int tg = t0 - 4 + j, thh = t0 - Hh + j;
      vg[jj] = (j < TT + 8 && tg >= 0 && tg < T) ? tg * 4 : OOB;
      vh[jj] = (j < TT + 2 * Hh && thh >= 0 && thh < T) ? thh * 4 : OOB;
    }
    const int sbC = b * C * T * 4, sbN = b * NARROW * T * 4;
#pragma unroll
    for (int q = 0; q < QX; ++q) {
      const int r = wave + NW * q;                    // rows >= C read the next frame's data: never stored
      const int so = sbC + r * T * 4;
      if (PLR) rx[q] = bl(sx, vt, (b * Cx + r) * T * 4);
      if (P9) ry[q] = bl(sy, vt, so);
    }
    if (PLR) {
#pragma unroll
      for (int q = 0; q < QA; ++q) ra[q] = bl(sa, vt, 2 * sbN + (wave + NW * q) * T * 4);
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int so = sbN + (wave + NW * q) * T * 4;   // rows >= 20: never stored
      if (PLR) rz[q] = bl(sz, vt, so);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        if (P9) rg[q][jj] = bl(sg, vg[jj], so);
        if (PLR) rh[q][jj] = bl(sh, vh[jj], so);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int q = 0; q < QX; ++q) {
      const int r = wave + NW * q;
      if (PLR && r < Cx) xn[r * ldn + lane] = rx[q];
      if (P9 && r < C) dys[r * ldn + lane] = ry[q];
    }
    if (PLR) {
#pragma unroll
      for (int q = 0; q < QA; ++q) dl[(wave + NW * q) * ldn + lane] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int r = wave + NW * q;
      if (r < NARROW) {
        if (PLR) dhs[r * ldn + lane] = rz[q];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = lane + 64 * jj;
          if (P9 && j < ldg) gs[r * ldg + j] = rg[q][jj];
          if (PLR && j < ldh) hs[r * ldh + j] = rh[q][jj];
        }
      }
    }
  };

  if (wg < a.ntiles) load_tile(wg);
  for (int tile = wg; tile < a.ntiles; tile += nwg) {
    const int b = tile / a.tiles_per_frame;
    const int t0 = (tile - b * a.tiles_per_frame) * TT;
    __syncthreads();                       // everyone is done reading the previous tile
    if (!(a.skip & 16)) store_tile();
    __syncthreads();
    if (tile + nwg < a.ntiles && !(a.skip & 8)) load_tile(tile + nwg);   // in flight during the MFMA loop below
    if (!(a.skip & 2))
#pragma unroll NSC_WG_UNROLL
    for (int s = 0; s < TT / 4; ++s) {
      const int tl = 4 * s + kq;
      if constexpr (P9) {
        float af[R9], bf[RT9];
#pragma unroll
        for (int r = 0; r < R9; ++r) af[r] = gs[off9[r] + tl];
#pragma unroll
        for (int c = 0; c < RT9; ++c) bf[c] = sm[offb9[c] + tl];
#pragma unroll
        for (int r = 0; r < R9; ++r)
#pragma unroll
          for (int c = 0; c < RT9; ++c) g9[r][c] = mfma4(af[r], bf[c], g9[r][c]);
      }
      if constexpr (PLR) {
        float af[RLR], bf[3];
#pragma unroll
        for (int r = 0; r < RLR; ++r) af[r] = hs[offlr[r] + tl];
#pragma unroll
        for (int c = 0; c < 3; ++c) bf[c] = sm[offb_lr[c] + tl];
#pragma unroll
        for (int r = 0; r < RLR; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) glr[r][c] = mfma4(af[r], bf[c], glr[r][c]);
      }
      if constexpr (PLR) {
        const float b0 = dhs[l15 * ldn + tl];
        const float b1 = sm[offb1 + tl];
#pragma unroll
        for (int r = 0; r < R1; ++r) {
          const float af1 = xn[off1[r] + tl];
          g1[r][0] = mfma4(af1, b0, g1[r][0]);
          g1[r][1] = mfma4(af1, b1, g1[r][1]);
        }
      }
    }
    // fused 1x1 data gradient AFTER the MFMA loop: by now the next tile's prefetch has landed, so these weight loads do
    // not queue behind it (vmcnt retires in order), and their registers are not live during the loop above.
    constexpr int RH = NW == 8 ? (RT9 + 1) / 2 : RT9;  // row tiles per wave in D1 (8 waves: halves [0,RH) and [RH,2RH))
    const int rbase = (wave >> 2) * RH;
    float av[5][RH];
    if (PART == 0 && a.dx) {
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int r = 0; r < RH; ++r) {
          const int c = (rbase + r) * 16 + l15;
          av[s][r] = a.wt1[(s * 4 + kq) * C + (c < C ? c : C - 1)];
        }
    }
    if (PART == 0 && a.dx && !(a.skip & 1)) {
      // fused 1x1 data gradient: dx = (W1^T dz1 + dy) * act'(x); wave owns column tile `wave`, all RT9 row tiles, K = 20
      f32x4 acc[RH];
#pragma unroll
      for (int r = 0; r < RH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int tt = (wave & 3) * 16 + l15;
#pragma unroll
      for (int s = 0; s < 5; ++s) {
        const float bv = dhs[(s * 4 + kq) * ldn + tt];
#pragma unroll
        for (int r = 0; r < RH; ++r) acc[r] = mfma4(av[s][r], bv, acc[r]);
      }
      const int t = t0 + tt;
      if (t < T) {
#pragma unroll
        for (int r = 0; r < RH; ++r)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int c = (rbase + r) * 16 + kq * 4 + reg;
            if (c < C) {
              float v = acc[r][reg] + dys[c * ldn + tt];
              if (a.in_act == NSC_ACT_LRELU) v *= (xn[c * ldn + tt] > 0.f ? 1.f : NSC_LRELU_ALPHA);
              a.dx[((long)b * C + c) * T + t] = v;
            }
          }
      }
    }
  }

  // ---- flush: either float atomics straight into the gradients, or plain coalesced stores into this workgroup's
  // private slab (summed afterwards by slab_reduce_kernel): 256 workgroups x 32k atomics on the same addresses cost
  // ~100 us per block, the store + reduce path ~15 us.
  const bool plain = a.slab_stride != 0;
  const long so = (long)slab_id * a.slab_stride;
  if (a.skip & 4) return;
  if constexpr (P9)
#pragma unroll
  for (int r = 0; r < R9; ++r)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int kk = (wave + NW * r) * 16 + kq * 4 + reg;
      if (kk > K9 * NARROW) continue;
#pragma unroll
      for (int cc = 0; cc < RT9; ++cc) {
        const int o = cc * 16 + l15;
        if (o >= C) continue;
        if (kk < K9 * NARROW) wg_flush(a.dw9 + so + (long)kk * C + o, g9[r][cc][reg], plain);
        else wg_flush(a.db9 + so + o, g9[r][cc][reg], plain);
      }
    }
  if constexpr (PLR)
#pragma unroll
  for (int r = 0; r < RLR; ++r)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int kk = (wave + NW * r) * 16 + kq * 4 + reg;
      if (kk > K15 * NARROW) continue;
#pragma unroll
      for (int ct = 0; ct < 3; ++ct) {
        const int c = ct * 8 + (l15 >> 2) * 2 + (l15 & 1);
        if (c >= NARROW) continue;
        const bool gate = (l15 & 2) != 0;
        if (kk < K15 * NARROW) wg_flush((gate ? a.dwr : a.dwl) + so + kk * NARROW + c, glr[r][ct][reg], plain);
        else wg_flush((gate ? a.dbr : a.dbl) + so + c, glr[r][ct][reg], plain);
      }
    }
  if constexpr (PLR)
#pragma unroll
  for (int r = 0; r < R1; ++r)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ci = (wave + NW * r) * 16 + kq * 4 + reg;
      if (ci > Cx) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int o = c * 16 + l15;
        if (o >= NARROW) continue;
        if (ci < Cx) wg_flush(a.dw1 + so + ci * NARROW + o, g1[r][c][reg], plain);
        else wg_flush(a.db1 + so + o, g1[r][c][reg], plain);
      }
    }
}

template <int RT9, int NW, int PART>
__global__ __launch_bounds__(64 * NW, 2) void gated_block_wgrad_kernel(BlockWgradArgs a, int ldn, int ldg, int ldh) {
  block_wgrad_body<RT9, NW, PART>(a, ldn, ldg, ldh, blockIdx.x, gridDim.x, blockIdx.x);
}

// ---- batched form: ONE launch produces the parameter gradients of up to NSC_WG_MAXJ gated blocks.  The weight
// gradients are off the critical path (only the optimizer reads them), so the engine defers them to the end of the
// backward pass; a (block, part) pair then gets ~256/njobs persistent workgroups that each walk 20-60 tiles instead
// of 2-4, which amortises the prologue, the register-resident accumulators' flush and the slab reduction (14x fewer
// slabs per block), and removes 4 launches per block.  Workgroup w: part = 1 + (w & 1) (two 4-wave workgroups of
// different parts share a CU); among the workgroups of a part, job j owns [wg0[j], wg0[j+1]).
template <int RT9>
__global__ __launch_bounds__(256, 2) void gated_block_wgrad_batch_kernel(BlockWgradBatch t, int ldn, int ldg) {
  const int part = 1 + (blockIdx.x & 1), w = blockIdx.x >> 1;
  int j = 0;
  while (j + 1 < t.njobs && w >= t.wg0[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const BlockWgradArgs& a = t.a[j];
  int ldh = 64 + 14 * a.dil;
  while ((ldh & 31) != 2) ++ldh;
  const int wl = w - t.wg0[j], nw = t.wg0[j + 1] - t.wg0[j];
  if (part == 1) block_wgrad_body<RT9, 4, 1>(a, ldn, ldg, ldh, wl, nw, w);
  else block_wgrad_body<RT9, 4, 2>(a, ldn, ldg, ldh, wl, nw, w);
}

// per job: grads[range] += sum of its workgroups' slabs; blockIdx.y = job * 2 + half (half 0: everything before dW9 -
// written by the part-2 workgroups; half 1: dW9 | db9 - written by the part-1 workgroups; both use slabs wg0[j]..wg0[j+1])
struct SlabReduceBatch {                         // jobs of BOTH block widths of a step: one launch reduces them all
  float* grads[2 * NSC_WG_MAXJ];
  const float* slab[2 * NSC_WG_MAXJ];            // slab 0 of the job's launch
  long stride[2 * NSC_WG_MAXJ];
  int off9[2 * NSC_WG_MAXJ], range[2 * NSC_WG_MAXJ];
  int w0[2 * NSC_WG_MAXJ], w1[2 * NSC_WG_MAXJ];  // the job's workgroup slabs [w0, w1)
};
__global__ void slab_reduce_batch_kernel(SlabReduceBatch t) {
  const int j = blockIdx.y >> 1, half = blockIdx.y & 1;
  const int i0 = half ? t.off9[j] : 0, i1 = half ? t.range[j] : t.off9[j];
  const int w0 = t.w0[j], w1 = t.w1[j];
  const float* __restrict__ slab = t.slab[j];
  const long stride = t.stride[j];
  float* g = t.grads[j];
  for (int i = i0 + blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s0 += slab[(long)w * stride + i];
      s1 += slab[(long)(w + 1) * stride + i];
      s2 += slab[(long)(w + 2) * stride + i];
      s3 += slab[(long)(w + 3) * stride + i];
    }
    for (; w < w1; ++w) s0 += slab[(long)w * stride + i];
    g[i] += (s0 + s1) + (s2 + s3);      // one adder per element: no atomics needed
  }
}

// The same for a launch with FEW jobs (a caller that flushes block by block: the op surface's BlockFn): the kernel above would run 64
// workgroups per half, each thread walking all 256 slabs - latency-bound (40 us for one C = 100 job).  Here a workgroup owns 64
// elements; its four waves take every fourth slab (four loads in flight each), partial sums meet in LDS, still one adder per element.
__global__ __launch_bounds__(256) void slab_reduce_batch_wide_kernel(SlabReduceBatch t) {
  __shared__ float part[4][64];
  const int j = blockIdx.y >> 1, half = blockIdx.y & 1;
  const int i0 = half ? t.off9[j] : 0, i1 = half ? t.range[j] : t.off9[j];
  const int w0 = t.w0[j], w1 = t.w1[j];
  const float* __restrict__ slab = t.slab[j];
  const long stride = t.stride[j];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = i0 + blockIdx.x * 64; base < i1; base += gridDim.x * 64) {   // uniform per workgroup
    const int i = base + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < i1) {
      int w = w0 + wave;
      for (; w + 12 < w1; w += 16) {
        s0 += slab[(long)w * stride + i];
        s1 += slab[(long)(w + 4) * stride + i];
        s2 += slab[(long)(w + 8) * stride + i];
        s3 += slab[(long)(w + 12) * stride + i];
      }
      for (; w < w1; w += 4) s0 += slab[(long)w * stride + i];
    }
    part[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0 && i < i1) t.grads[j][i] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    __syncthreads();
  }
}

// grads[i] += sum_w slab[w * stride + i]; blockIdx.y splits the slabs (8 groups) so enough loads are in flight
__global__ void slab_reduce_kernel(const float* __restrict__ slab, long stride, int nslabs, float* __restrict__ grads, int n) {
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  const int w0 = blockIdx.y * per, w1 = min(nslabs, w0 + per);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = w0;
    for (; w + 3 < w1; w += 4) {
      s0 += slab[(long)w * stride + i];
      s1 += slab[(long)(w + 1) * stride + i];
      s2 += slab[(long)(w + 2) * stride + i];
      s3 += slab[(long)(w + 3) * stride + i];
    }
    for (; w < w1; ++w) s0 += slab[(long)w * stride + i];
    atomicAdd(grads + i, (s0 + s1) + (s2 + s3));
  }
}

static int ld2(int w) { int l = w; while ((l & 31) != 2) ++l; return l; }

extern "C" long nsc_gated_block_wgrad_workspace(int C) {
  // floats needed for the slab flush: 256 workgroups x (the block's contiguous parameter range, padded)
  const long range = (long)C * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW) + (long)K9 * NARROW * C + C;
  return 256L * ((range + 63) & ~63L);
}

extern "C" int nsc_gated_block_wgrad(const float* x, const float* h, const float* g, const float* dy, const float* da,
                                     const float* dz1, float* dw1, float* db1, float* dwl, float* dbl, float* dwr,
                                     float* dbr, float* dw9, float* db9, const float* wt1, float* dx, int in_act, int B,
                                     int C, int T, int narrow, int k9, int dil, int waves, int part, float* workspace,
                                     void* stream) {
  NSC_REQUIRE(x && h && g && dy && da && dz1 && dw1 && db1 && dwl && dbl && dwr && dbr && dw9 && db9,
              NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: null pointer");
  NSC_REQUIRE(!dx || wt1, NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: dx needs wt1");
  NSC_REQUIRE(in_act == NSC_ACT_NONE || in_act == NSC_ACT_LRELU, NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: in_act must be none|lrelu");
  NSC_REQUIRE(B > 0 && C > 1 && T > 0 && dil > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: bad sizes");
  NSC_REQUIRE(narrow == NARROW && k9 == K9 && dil <= 4 && C <= 112, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_wgrad: built for narrow=20, k9=9, dil<=4, C<=112 (got %d, %d, %d, %d)", narrow, k9, dil, C);
  NSC_REQUIRE(part >= 0 && part <= 2 && !(part && dx), NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: part in 0..2; dx needs part 0");
  const int ldn = ld2(64), ldg = ld2(72), ldh = ld2(64 + 14 * dil);
  const size_t fl = (size_t)(C + 2) * ldn + (part != 2 ? (size_t)C * ldn + (size_t)(NARROW + 2) * ldg : 0) +
                    (part != 1 ? (size_t)(NARROW + 2) * ldh : 0) + (size_t)3 * NARROW * ldn;
  const size_t smem = fl * sizeof(float);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "nsc_gated_block_wgrad: %zu B LDS", smem);
  BlockWgradArgs a{B, C, T, dil, x, h, g, dy, da, dz1, wt1, dx, in_act, dw1, db1, dwl, dbl, dwr, dbr, dw9, db9, 0, 0, 0, 0, C};
  { static int skip_env = -1; if (skip_env < 0) { skip_env = NSC_PROBE_INT("NSC_WG_SKIP", 0); } a.skip = skip_env; }
  a.tiles_per_frame = nsc_cdiv(T, 64);
  a.ntiles = B * a.tiles_per_frame;
  const int grid = std::min(a.ntiles, 256);
  // slab flush needs the eight gradients to be one contiguous range in creation order (they are, in the flat buffer)
  const long range = (long)C * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW) + (long)K9 * NARROW * C + C;
  const bool contiguous = db1 == dw1 + (long)C * NARROW && dwl == db1 + NARROW && dbl == dwl + K15 * NARROW * NARROW &&
                          dwr == dbl + NARROW && dbr == dwr + K15 * NARROW * NARROW && dw9 == dbr + NARROW &&
                          db9 == dw9 + (long)K9 * NARROW * C;
  const bool use_slab = workspace && contiguous && grid > 8;
  float* gbase = dw1;
  if (use_slab) {
    a.slab_stride = (range + 63) & ~63L;
    const long off = workspace - dw1;   // redirect every gradient pointer into slab 0, keeping the relative layout
    a.dw1 += off; a.db1 += off; a.dwl += off; a.dbl += off; a.dwr += off; a.dbr += off; a.dw9 += off; a.db9 += off;
  }
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_WG(RT, NW_)                                                                                          \
  do {                                                                                                              \
    auto kern = part == 0 ? gated_block_wgrad_kernel<RT, NW_, 0>                                                    \
                          : (part == 1 ? gated_block_wgrad_kernel<RT, NW_, 1> : gated_block_wgrad_kernel<RT, NW_, 2>); \
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);   \
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_wgrad: smem attr: %s", hipGetErrorString(e));         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW_), smem, st, a, ldn, ldg, ldh);                               \
  } while (0)
  NSC_REQUIRE(waves == 4 || waves == 8, NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad: waves must be 4 or 8");
  if (nsc_cdiv(C, 16) <= 4) { if (waves == 8) LAUNCH_WG(4, 8); else LAUNCH_WG(4, 4); }
  else { if (waves == 8) LAUNCH_WG(7, 8); else LAUNCH_WG(7, 4); }
#undef LAUNCH_WG
  NSC_CHECK_LAUNCH("gated_block_wgrad");
  if (use_slab) {
    // reduce only the sub-range this launch wrote (part 1: dW9 | db9 at the end; part 2: everything before it)
    const long off9 = (long)C * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW);
    const long r0 = part == 1 ? off9 : 0, r1 = part == 2 ? off9 : range;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(nsc_cdiv(r1 - r0, 256), 8), dim3(256), 0, st, workspace + r0, a.slab_stride,
                       grid, gbase + r0, (int)(r1 - r0));
    NSC_CHECK_LAUNCH("slab_reduce");
  }
  return NSC_OK;
}

template <int RT9>
static int launch_wgrad_batch(const nsc_block_wgrad_job* jobs, const int* idx, int n, int B, float* workspace,
                              long workspace_floats, hipStream_t st, SlabReduceBatch& r, int& nr, long& used_floats, bool split) {
  BlockWgradBatch t;
  memset(&t, 0, sizeof(t));
  static int skip_env = -1;
  if (skip_env < 0) { skip_env = NSC_PROBE_INT("NSC_WG_SKIP", 0); }
  const int ldn = ld2(64), ldg = ld2(72);
  long stride = 0, total_tiles = 0;
  size_t smem = 0;
  for (int q = 0; q < n; ++q) {
    const nsc_block_wgrad_job& jb = jobs[idx[q]];
    const int C = jb.C, Cx = jb.Cin > 0 ? jb.Cin : jb.C;
    const long range = (long)Cx * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW) + (long)K9 * NARROW * C + C;
    stride = std::max(stride, (range + 63) & ~63L);
    BlockWgradArgs& a = t.a[q];
    a.B = B; a.C = C; a.Cin = Cx; a.T = jb.T; a.dil = jb.dil;
    a.x = jb.x; a.h = jb.h; a.g = jb.g; a.dy = jb.dy; a.da = jb.da; a.dz1 = jb.dz1;
    a.wt1 = nullptr; a.dx = nullptr; a.in_act = 0;
    // gradient pointers redirected into slab 0 of the workspace, keeping the block's relative layout
    float* p = workspace;
    a.dw1 = p; p += (long)Cx * NARROW; a.db1 = p; p += NARROW;
    a.dwl = p; p += K15 * NARROW * NARROW; a.dbl = p; p += NARROW;
    a.dwr = p; p += K15 * NARROW * NARROW; a.dbr = p; p += NARROW;
    a.dw9 = p; p += (long)K9 * NARROW * C; a.db9 = p;
    a.tiles_per_frame = nsc_cdiv(jb.T, 64);
    a.ntiles = B * a.tiles_per_frame;
    a.skip = skip_env;
    total_tiles += a.ntiles;
    r.grads[nr + q] = jb.grads;
    r.off9[nr + q] = (int)((long)Cx * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW));
    r.range[nr + q] = (int)range;
    const int ldh = ld2(64 + 14 * jb.dil);
    const size_t f1 = (size_t)(C + 2) * ldn + (size_t)C * ldn + (size_t)(NARROW + 2) * ldg + (size_t)3 * NARROW * ldn;
    const size_t f2 = (size_t)(C + 2) * ldn + (size_t)(NARROW + 2) * ldh + (size_t)3 * NARROW * ldn;
    smem = std::max(smem, std::max(f1, f2) * sizeof(float));
  }
  NSC_REQUIRE(smem <= 80 * 1024, NSC_ERR_UNSUPPORTED, "nsc_gated_block_wgrad_batch: %zu B LDS", smem);
  // workgroup slots per job ~ its share of the tiles (largest remainder, at least one each, 256 in all)
  const int slots = 256;
  NSC_REQUIRE((long)slots * stride <= workspace_floats, NSC_ERR_BAD_ARG,
              "nsc_gated_block_wgrad_batch: workspace %ld floats < %ld", workspace_floats, (long)slots * stride);
  int cnt[NSC_WG_MAXJ], used = 0;
  for (int q = 0; q < n; ++q) {
    cnt[q] = std::max(1, (int)((long)slots * t.a[q].ntiles / total_tiles));
    cnt[q] = std::min(cnt[q], t.a[q].ntiles);
    used += cnt[q];
  }
  for (int guard = 0; used < slots && guard < 4 * slots; ++guard) {   // hand out the rest to the most loaded jobs
    int best = -1;
    double bl = 0;
    for (int q = 0; q < n; ++q) {
      const double l = (double)t.a[q].ntiles / cnt[q];
      if (cnt[q] < t.a[q].ntiles && l > bl) { bl = l; best = q; }
    }
    if (best < 0) break;
    ++cnt[best];
    ++used;
  }
  while (used > slots) {                                              // (only if njobs > slots; cannot happen for MAXJ)
    int best = 0;
    for (int q = 1; q < n; ++q) if (cnt[q] > cnt[best]) best = q;
    --cnt[best];
    --used;
  }
  t.wg0[0] = 0;
  for (int q = 0; q < n; ++q) t.wg0[q + 1] = t.wg0[q] + cnt[q];
  for (int q = 0; q < n; ++q) {
    r.w0[nr + q] = t.wg0[q];
    r.w1[nr + q] = t.wg0[q + 1];
    r.slab[nr + q] = workspace;
    r.stride[nr + q] = stride;
  }
  nr += n;
  used_floats = (long)slots * stride;
  t.njobs = n;
  for (int q = 0; q < n; ++q) t.a[q].slab_stride = stride;
  // split operands on the bf16 matrix cores (block_split.hip): same job table, slabs and flush layout
  if (split && nsc_block_wgrad_split_ok(t)) return nsc_launch_block_wgrad_split(t, RT9, used, st);
  auto kern = gated_block_wgrad_batch_kernel<RT9>;
  const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_wgrad_batch: smem attr: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(2 * used), dim3(256), smem, st, t, ldn, ldg);
  NSC_CHECK_LAUNCH("gated_block_wgrad_batch");
  return NSC_OK;
}
static int launch_slab_reduce(SlabReduceBatch& r, int& nr, hipStream_t st) {
  if (nr == 0) return NSC_OK;
  if (nr <= 2) {
    hipLaunchKernelGGL(slab_reduce_batch_wide_kernel, dim3(512, 2 * nr), dim3(256), 0, st, r);
    NSC_CHECK_LAUNCH("slab_reduce_batch_wide");
    nr = 0;
    return NSC_OK;
  }
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3(64, 2 * nr), dim3(256), 0, st, r);
  NSC_CHECK_LAUNCH("slab_reduce_batch");
  nr = 0;
  return NSC_OK;
}

extern "C" long nsc_gated_block_wgrad_batch_workspace(int Cmax) { return nsc_gated_block_wgrad_workspace(Cmax); }

static int wgrad_batch_impl(const nsc_block_wgrad_job* jobs, int njobs, int B, int narrow, int k9, float* workspace,
                            long workspace_floats, void* stream, bool split) {
  NSC_REQUIRE(jobs && njobs > 0 && B > 0 && workspace, NSC_ERR_BAD_ARG, "nsc_gated_block_wgrad_batch: bad arguments");
  NSC_REQUIRE(narrow == NARROW && k9 == K9, NSC_ERR_UNSUPPORTED, "nsc_gated_block_wgrad_batch: built for narrow=20, k9=9");
  int small[NSC_WG_MAXJ], big[NSC_WG_MAXJ], ns = 0, nb = 0;
  hipStream_t st = (hipStream_t)stream;
  // One launch per block width, then ONE slab reduce for both - if the workspace holds both launches' slabs side by side (the
  // engine's does: 2 x nsc_gated_block_wgrad_batch_workspace); with a smaller one the second launch reuses the slabs of the first
  // after its reduce (two reduce launches, as before round 4).
  SlabReduceBatch r;
  memset(&r, 0, sizeof(r));
  int nr = 0;
  long ws_used = 0;                    // floats of the workspace that hold slabs not reduced yet
  auto one = [&](auto rt_c, const int* idx, int& n) -> int {
    constexpr int RT = decltype(rt_c)::value;
    long need = 0;
    for (int q = 0; q < n; ++q) {      // what launch_wgrad_batch will take: 256 slabs of the widest job's parameter range
      const nsc_block_wgrad_job& jb = jobs[idx[q]];
      const long Cx = jb.Cin > 0 ? jb.Cin : jb.C;
      const long range = Cx * NARROW + NARROW + 2L * (K15 * NARROW * NARROW + NARROW) + (long)K9 * NARROW * jb.C + jb.C;
      need = std::max(need, 256L * ((range + 63) & ~63L));
    }
    if (nr && (ws_used + need > workspace_floats || nr + n > 2 * NSC_WG_MAXJ)) {
      int rc = launch_slab_reduce(r, nr, st);
      if (rc) return rc;
      ws_used = 0;
    }
    long used = 0;
    int rc = launch_wgrad_batch<RT>(jobs, idx, n, B, workspace + ws_used, workspace_floats - ws_used, st, r, nr, used, split);
    if (rc) return rc;
    ws_used += used;
    n = 0;
    return NSC_OK;
  };
  auto flush = [&](bool all) -> int {
    if (ns && (all || ns == NSC_WG_MAXJ)) {
      int rc = one(std::integral_constant<int, 4>{}, small, ns);
      if (rc) return rc;
    }
    if (nb && (all || nb == NSC_WG_MAXJ)) {
      int rc = one(std::integral_constant<int, 7>{}, big, nb);
      if (rc) return rc;
    }
    return all ? launch_slab_reduce(r, nr, st) : NSC_OK;
  };
  for (int j = 0; j < njobs; ++j) {
    const nsc_block_wgrad_job& jb = jobs[j];
    NSC_REQUIRE(jb.x && jb.h && jb.g && jb.dy && jb.da && jb.dz1 && jb.grads, NSC_ERR_BAD_ARG,
                "nsc_gated_block_wgrad_batch: job %d has a null pointer", j);
    NSC_REQUIRE(jb.C > 1 && jb.C <= 112 && jb.T > 0 && jb.dil > 0 && jb.dil <= 4 && (jb.Cin == 0 || jb.Cin == 1 || jb.Cin == jb.C),
                NSC_ERR_UNSUPPORTED, "nsc_gated_block_wgrad_batch: job %d: C %d, Cin %d, T %d, dil %d unsupported", j, jb.C, jb.Cin,
                jb.T, jb.dil);
    if (nsc_cdiv(jb.C, 16) <= 4) small[ns++] = j; else big[nb++] = j;
    int rc = flush(false);
    if (rc) return rc;
  }
  return flush(true);
}
extern "C" int nsc_gated_block_wgrad_batch(const nsc_block_wgrad_job* jobs, int njobs, int B, int narrow, int k9,
                                           float* workspace, long workspace_floats, void* stream) {
  return wgrad_batch_impl(jobs, njobs, B, narrow, k9, workspace, workspace_floats, stream, false);
}
// the same on the bf16 matrix cores with split operands (block_split.hip); launches whose shapes it does not serve (T % 4 != 0,
// unaligned tensors, dilation > 2) run the exact kernel
extern "C" int nsc_gated_block_wgrad_batch_split(const nsc_block_wgrad_job* jobs, int njobs, int B, int narrow, int k9,
                                                 float* workspace, long workspace_floats, void* stream) {
  return wgrad_batch_impl(jobs, njobs, B, narrow, k9, workspace, workspace_floats, stream, true);
}


// =====================================================================================================
// Fused DATA-PATH backward of the gated block (8 waves, ~78 KB LDS -> two workgroups per CU):
//   dg    = W9^T * dy                         (k9 data gradient; split over taps of both parities -> 24 half-jobs)
//   dlin  = dg.th ; dgate = dg.lin.(1-th^2)   (GLU backward, in LDS)                       -> da [B,40,T]
//   dz1   = (Wl^T dlin + Wr^T dgate).lrelu'(h)(both k15 data gradients in one pass)        -> dz1 [B,20,T]
//   dx    = (W1^T dz1 + dy).act'(x)           (1x1 data gradient + residual)               -> dx [B,C,T]
// Inputs are the activations the forward saved (h, lin, th) and dy; the three outputs are exactly what the
// persistent weight-gradient kernel consumes.  Replaces four launches per block on the critical stream.
// Columns: dys j <-> t0-Hh-4+j ; lin/th/dg j <-> t0-Hh+j ; dhs j <-> t0+j   (Hh = 7*dil).
// =====================================================================================================

template <int RT9>
__global__ __launch_bounds__(512) void gated_block_dgrad_kernel(BlockDgradArgs a, int ldy, int lda, int ldn) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64;
  const int C = a.C, T = a.T, d = a.dil;
  const int Hh = 7 * d;
  const int W_a = TT + 2 * Hh, W_dy = W_a + 8;
  const int C4 = (C + 3) & ~3;
  const int nct_a = (W_a + 15) >> 4;                 // 6 (d=2) / 5 (d=1)
  float* dys = sm;                                   // [C4][ldy]   (pad rows zero)
  float* lin = dys + C4 * ldy;                       // [NARROW][lda]  -> dlin
  float* th = lin + NARROW * lda;                    // [NARROW][lda]  -> dgate
  float* dg = th + NARROW * lda;                     // [NARROW][lda]
  float* dhs = dg + NARROW * lda;                    // [NARROW][ldn]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;

  // ---- stage dy (with halo), lin, th; zero dg ----
  nsc_stage_rows<8>(dys, ldy, C4, C, W_dy, a.dy + (long)b * C * T, T, t0 - Hh - 4, T, 0, wave, lane);
  nsc_stage_rows<8>(lin, lda, NARROW, NARROW, W_a, a.lin + (long)b * NARROW * T, T, t0 - Hh, T, 0, wave, lane);
  nsc_stage_rows<8>(th, lda, NARROW, NARROW, W_a, a.th + (long)b * NARROW * T, T, t0 - Hh, T, 0, wave, lane);
  for (int e = tid; e < NARROW * lda; e += 512) dg[e] = 0.f;
  __syncthreads();

  // ---- D9: dg[ci][ja] = sum_{tap', o} wt9[tap'][o][ci] * dy[o][ja + tap'] ----
  // job = (row tile rt, column tile ct), split in two halves over the parity of tap' -> half-job hj = 2*job + parity.
  // wave w runs half-jobs 3w .. 3w+2 (2*nct_a*2 = 24 at d=2; at d=1 the last waves run discarded duplicates).
  {
    const int nhj = 2 * nct_a * 2;
    int rt[3], par[3], jcol[3];
    bool real[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      int hj = 3 * wave + e;
      real[e] = hj < nhj;
      hj = real[e] ? hj : 0;
      par[e] = hj & 1;
      const int job = hj >> 1;
      rt[e] = job & 1;
      jcol[e] = (job >> 1) * 16 + l15;
    }
    const __amdgpu_buffer_rsrc_t wsrd =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt9), 0, K9 * C * NARROW * 4, 0x00020000);
    int voff[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int ci = rt[e] * 16 + l15;
      voff[e] = (kq * NARROW + (ci < NARROW ? ci : NARROW - 1)) * 4;   // rows >= 20: clamped, never stored
    }
    const int ncq = C4 >> 2;
    const int tap_bytes = C * NARROW * 4, step_bytes = 4 * NARROW * 4;
    f32x4 acc[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int G9 = 5;
    float an[G9][3];
    int ip = 0, cp = 0;                               // prefetch cursor: tap slot i (tap' = parity + 2 i), channel group
    auto fetch9 = [&]() {
#pragma unroll
      for (int u = 0; u < G9; ++u) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          // tap' >= 9 (fifth slot of the odd parity, or past the end) lands outside the descriptor -> 0
          const int soff = __builtin_amdgcn_readfirstlane((par[e] + 2 * ip) * tap_bytes + cp * step_bytes);
          an[u][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wsrd, voff[e], soff, 0));
        }
        const bool wr = (cp + 1 == ncq);
        cp = wr ? 0 : cp + 1;
        ip += wr ? 1 : 0;
      }
    };
    fetch9();
    const int nsteps = 5 * ncq;                        // 5 tap slots (even parity: taps 0,2,4,6,8; odd: 1,3,5,7,(9 -> zero))
    const int ngroups = (nsteps + G9 - 1) / G9;
    int it = 0, cq = 0;
    for (int g = 0; g < ngroups; ++g) {
      float ac[G9][3];
#pragma unroll
      for (int u = 0; u < G9; ++u)
#pragma unroll
        for (int e = 0; e < 3; ++e) ac[u][e] = an[u][e];
      fetch9();
#pragma unroll
      for (int u = 0; u < G9; ++u) {
        const int itc = it < 5 ? it : 4;
        const float* yrow = dys + (cq * 4 + kq) * ldy + 2 * itc;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const int tp = par[e] + ((par[e] && itc == 4) ? -1 : 0);   // keep the (zero-weight) tap 9 read inside the tile
          acc[e] = mfma4(ac[u][e], yrow[tp + jcol[e]], acc[e]);
        }
        const bool wr = (cq + 1 == ncq);
        cq = wr ? 0 : cq + 1;
        it += wr ? 1 : 0;
      }
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      if (!real[e]) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ci = rt[e] * 16 + kq * 4 + reg;
        if (ci < NARROW && jcol[e] < lda) atomicAdd(&dg[ci * lda + jcol[e]], acc[e][reg]);
      }
    }
  }
  __syncthreads();
  // ---- GLU backward in place (lin/th are zero outside the frame, so dlin/dgate are too); da -> global ----
  for (int e = tid; e < NARROW * lda; e += 512) {
    const float l = lin[e], tg = th[e], gg = dg[e];
    const float dl_ = gg * tg, dgt = gg * l * (1.f - tg * tg);
    lin[e] = dl_;
    th[e] = dgt;
    const int c = e / lda, ja = e - c * lda;
    const int t = t0 - Hh + ja;
    if (ja >= Hh && ja < Hh + TT && t < T) {
      a.da[((long)b * 2 * NARROW + c) * T + t] = dl_;
      a.da[((long)b * 2 * NARROW + NARROW + c) * T + t] = dgt;
    }
  }
  __syncthreads();

  // ---- D15: dz1[ci][tt] = (sum_{tap',br,c} wt{l,r}[tap'][c][ci] d{lin,gate}[c][tt + tap' d]) * lrelu'(h); one job per wave ----
  {
    const int rt = wave & 1, ct = wave >> 1;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int tt = ct * 16 + l15;
    const int ci = rt * 16 + l15;
    const int cic = ci < NARROW ? ci : NARROW - 1;
    constexpr int G15 = 10;                             // one tap = 2 branches x 5 k-steps
    float an[G15];
    auto fetch15 = [&](int tapf) {
      const int tf = tapf < K15 ? tapf : K15 - 1;
#pragma unroll
      for (int u = 0; u < G15; ++u) {
        const float* wsrc = (u < 5) ? a.wtl : a.wtr;
        an[u] = wsrc[(tf * NARROW + (u % 5) * 4 + kq) * NARROW + cic];
      }
    };
    fetch15(0);
    for (int tap = 0; tap < K15; ++tap) {
      float ac[G15];
#pragma unroll
      for (int u = 0; u < G15; ++u) ac[u] = an[u];
      fetch15(tap + 1);
#pragma unroll
      for (int u = 0; u < G15; ++u) {
        const float* src = (u < 5) ? lin : th;
        acc = mfma4(ac[u], src[((u % 5) * 4 + kq) * lda + tt + tap * d], acc);
      }
    }
    const int t = t0 + tt;
    const bool live = t < T;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int c = rt * 16 + kq * 4 + reg;
      if (c < NARROW) {
        const float hv = live ? a.h[((long)b * NARROW + c) * T + t] : 0.f;
        const float v = live ? acc[reg] * (hv > 0.f ? 1.f : NSC_LRELU_ALPHA) : 0.f;
        dhs[c * ldn + tt] = v;
        if (live) a.dz1[((long)b * NARROW + c) * T + t] = v;
      }
    }
  }
  __syncthreads();

  // ---- D1: dx = (W1^T dz1 + dy) * act'(x); wave -> column tile (w & 3), row-tile half (w >> 2) ----
  {
    constexpr int RH = (RT9 + 1) / 2;
    const int rbase = (wave >> 2) * RH;
    const int tt = (wave & 3) * 16 + l15;
    f32x4 acc[RH];
    float av[5][RH];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int r = 0; r < RH; ++r) {
        const int c = (rbase + r) * 16 + l15;
        av[s][r] = a.wt1[(s * 4 + kq) * C + (c < C ? c : C - 1)];
      }
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float bv = dhs[(s * 4 + kq) * ldn + tt];
#pragma unroll
      for (int r = 0; r < RH; ++r) acc[r] = mfma4(av[s][r], bv, acc[r]);
    }
    const int t = t0 + tt;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < RH; ++r)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int c = (rbase + r) * 16 + kq * 4 + reg;
          if (c < C) {
            float v = acc[r][reg] + dys[c * ldy + tt + Hh + 4];
            const long gi = ((long)b * C + c) * T + t;
            if (a.in_act == NSC_ACT_LRELU) v *= (a.x[gi] > 0.f ? 1.f : NSC_LRELU_ALPHA);
            a.dx[gi] = v;
          }
        }
    }
  }
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>)
template <int N, int I = 0, class F>
__device__ __forceinline__ void nsc_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    nsc_static_for<N, I + 1>(f);
  }
}
#ifndef NSC_D9_NS
#define NSC_D9_NS 3
#endif
#ifndef NSC_D15_NS
#define NSC_D15_NS 3
#endif
// ---- helpers of gated_block_dgrad2_kernel's k9 data gradient ----
// rows 0..15 (channels 0..15): NC column tiles starting at yb, this wave's K-quarter (register-resident fragments)
// hook(std::integral_constant<int, st>) runs once per step, before the step's MFMAs (the caller's load-issue slot)
template <int NC, int NJ, int LDY_, class Hook>
__device__ __forceinline__ void d9_rows0(const float (&w9r)[K9][NJ], const float (&w9x)[3], const float* yb, const float* yx,
                                         int kg, f32x4 (&acc)[NC], Hook&& hook) {
  constexpr int NSTEP = K9 * NJ;
  // B fragments DEPTH - 1 steps ahead (compile-time slot index: no register copies).  A step of NC MFMAs covers 32 NC
  // cycles; an LDS round trip in this kernel is ~170 (the 4-tile loop ran at 42 cycles per MFMA with one step in flight),
  // so short steps keep more in flight: >= 8 MFMAs' worth.
  constexpr int DEPTH = NC >= 4 ? 3 : (NC == 3 ? 4 : 5);
  float bb[DEPTH][NC];
  auto fetch = [&](int stn) {
    const int tpn = stn / NJ, jn = stn - tpn * NJ;
#pragma unroll
#if defined(NSC_EXP) && (NSC_EXP & 4)
    for (int ct = 0; ct < NC; ++ct) bb[stn % DEPTH][ct] = w9x[ct % 3];     // timing experiment: no LDS operand reads
#else
    for (int ct = 0; ct < NC; ++ct) bb[stn % DEPTH][ct] = yb[16 * jn * LDY_ + tpn + ct * 16];
#endif
  };
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i) fetch(i);
  nsc_static_for<NSTEP>([&](auto st_c) {
    constexpr int st = decltype(st_c)::value;
    const int tp = st / NJ, j = st - tp * NJ;
    if (st + DEPTH - 1 < NSTEP) fetch(st + DEPTH - 1);
    hook(st_c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma4(w9r[tp][j], bb[st % DEPTH][ct], acc[ct]);
    __builtin_amdgcn_sched_barrier(0);
  });
  // the left-over channel group: this quarter's taps kg + 4 i (tap kg + 8 exists for quarter 0 only: wave-uniform)
  float bx[3][NC];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) bx[i][ct] = yx[4 * (i == 2 && kg != 0 ? 1 : i) + ct * 16];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i == 2 && kg != 0) break;
#pragma unroll
    for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma4(w9x[i], bx[i][ct], acc[ct]);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// channels 16..19, PACKED: a 16-row tile of [4 time shifts s][4 channels i] instead of 4 useful rows + 12 of padding.
// Row (s,i), column n  ->  dg[16+i][64 ctp + 4n + s] = sum_{m,o} A[(s,i)][(m,o)] dy[o][64 ctp + 4n + m],
// A = wt9[m-s][o][16+i] for 0 <= m-s < 9 else 0: 12 "taps" m instead of 9, but one MFMA column tile now spans 64 time steps
// instead of 16 (2 tiles instead of 6).  A from LDS (w9ps [9][C][4]), lane (4s+i, kq); B lanes walk time with stride 4.
// HALF: 0 / 1 = the first / second half of the (tap, half-group) steps (the two waves that share a K-quarter split them)
template <int NJ, int NK9_, int LDY_, int NT, int HALF>
__device__ __forceinline__ void d9_packed(const float* w9ps, int C, int w9t, const float* dys, int kg, int kq, int l15,
                                          f32x4 (&acc)[2]) {
  // NT = 2: acc[0] / acc[1] are the column tiles at +0 / +64.  NT = 1: one column tile; acc[0] / acc[1] take alternate
  // k-steps (independent MFMA chains) and the caller adds them.
  const int sft = l15 >> 2, ich = l15 & 3;
  const float* ypb = dys + (4 * kg + kq) * LDY_ + 4 * l15;
  // The wave that runs this shares its SIMD with a wave streaming the dense k9 gradient from registers: whenever this stream
  // waits for an LDS operand the dense one takes the matrix pipe, and what is left over runs alone - latency-bound - at the
  // end of the phase (per-wave stamps: dense wave done after 9.8 k cycles, this one after 13.4 k).  So the operands are
  // requested TWO steps ahead (steps = half a tap's k-steps, three register slots).
  constexpr int NH = (NJ + 1) / 2, NSTEP_ALL = 2 * (K9 + 3), S0 = HALF * (NSTEP_ALL / 2), NSTEP = NSTEP_ALL / 2, NS = NSC_D9_NS;
  float av[NS][NH], b0[NS][NH], b1[NT == 2 ? NS : 1][NH];
  bool okv[NS];
#if defined(NSC_EXP) && (NSC_EXP & 8)
  float expv[2] = {w9ps[l15], ypb[0]};
#endif
  auto fetch = [&](int stl, int slot) {
    const int stn = S0 + stl;
    const int m = stn >> 1, j0 = (stn & 1) * NH;
    const int tap = m - sft;
    // 3 <= m <= 8: every shift 0..3 has a real tap (m is a compile-time constant of the unrolled step: the selects fold away)
    const bool ok = (m >= 3 && m <= K9 - 1) || (unsigned)tap < (unsigned)K9;
    okv[slot] = ok;
    const float* ap = w9ps + (ok ? tap : 0) * w9t + (4 * kg + kq) * 4 + ich;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      if (j0 + j < NJ) {
#if defined(NSC_EXP) && (NSC_EXP & 8)
        av[slot][j] = expv[0];
        b0[slot][j] = expv[1];
#else
        // (NK9 = 9 is the C = 25 shape staged as 36 rows: channel groups past C - their dy rows are zeros - would index the table
        // past its C entries per tap, and past its end for the last tap: whatever LDS holds there, NaN included.  Clamped.)
        av[slot][j] = NK9_ == 9 ? ap[(min(4 * kg + kq + 16 * (j0 + j), C - 1) - (4 * kg + kq)) * 4] : ap[16 * (j0 + j) * 4];
        b0[slot][j] = ypb[16 * (j0 + j) * LDY_ + m];
#endif
        if (NT == 2) b1[NT == 2 ? slot : 0][j] = ypb[16 * (j0 + j) * LDY_ + m + 64];
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) fetch(i, i);
#pragma unroll
  for (int st = 0; st < NSTEP; ++st) {
    if (st + NS - 1 < NSTEP) fetch(st + NS - 1, (st + NS - 1) % NS);
    __builtin_amdgcn_sched_barrier(0);
    const bool ok = okv[st % NS];
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int jj = ((S0 + st) & 1) * NH + j, m = (S0 + st) >> 1;
      if (jj < NJ) {
        const float a = (m >= 3 && m <= K9 - 1) || ok ? av[st % NS][j] : 0.f;
        if (NT == 2) {
          acc[0] = mfma4(a, b0[st % NS][j], acc[0]);
          acc[1] = mfma4(a, b1[NT == 2 ? st % NS : 0][j], acc[1]);
        } else {
          acc[(jj + m) & 1] = mfma4(a, b0[st % NS][j], acc[(jj + m) & 1]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // the left-over channel group (cq = NK9-1): this quarter's m = kg, kg+4, kg+8
  const float* ypx = dys + (4 * (NK9_ - 1) + kq) * LDY_ + 4 * l15 + kg;
#pragma unroll
  for (int e = HALF == 0 ? 0 : 2; e < (HALF == 0 ? 2 : 3); ++e) {      // first half: e = 0, 1; second half: e = 2
    const int tap = kg + 4 * e - sft;
    const bool ok = (unsigned)tap < (unsigned)K9;
    // channel rows >= C (C = 50: rows 50, 51) meet zero rows of the dy tile: clamp the index so the fragment is a finite
    // stand-in (an unclamped read ran past the table into uninitialised LDS: NaN * 0)
    const float a0 = w9ps[(ok ? tap : 0) * w9t + min(4 * (NK9_ - 1) + kq, C - 1) * 4 + ich];
    const float a = ok ? a0 : 0.f;
    if (NT == 2) {
      acc[0] = mfma4(a, ypx[4 * e], acc[0]);
      acc[1] = mfma4(a, ypx[4 * e + 64], acc[1]);
    } else {
      acc[e & 1] = mfma4(a, ypx[4 * e], acc[e & 1]);
    }
  }
}

// ---- helpers of the k15 data gradient (weights from LDS: w15s [15][40][20], tap stride W15T) ----
// W15T = 808: == 8 (mod 32), so the four time shifts of the packed tile (four different taps per lane group) read four
// different bank groups (at the natural stride 800 == 0 they were a 4-way conflict).
constexpr int W15T = 2 * NARROW * NARROW + 8;
// rows = channels 0..15, NC column tiles starting at ab; this wave's taps kg, kg+4, kg+8, (kg+12)
template <int NC, int DIL_, int LDA_, class Hook>
__device__ __forceinline__ void d15_rows0(const float* wb, const float* ab, int kg, f32x4 (&acc)[NC], Hook&& hook) {
  // groups of two k-steps, software-pipelined like d9_rows0: the operands of group g+1 are requested before the MFMAs of
  // group g issue (unpipelined, every group exposed a full LDS round trip: the phase ran at 2/3 of its MFMA rate)
  constexpr int NG = 4 * 5;                              // (tap quarter e4, channel-group pair u/2)
  constexpr int DEPTH = NC >= 4 ? 2 : 3;                 // groups in flight + 1 (a group = 2 NC MFMAs; see d9_rows0)
  float av[DEPTH][2], bv[DEPTH][2][NC];
#if defined(NSC_EXP) && (NSC_EXP & 4)
  float expv[2 + NC];
  for (int i = 0; i < 2 + NC; ++i) expv[i] = wb[i * 16];
#endif
  auto fetch = [&](int g) {
    const int e4 = g / 5, u = 2 * (g - 5 * e4), slot = g % DEPTH;
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
#if defined(NSC_EXP) && (NSC_EXP & 4)
      av[slot][uu] = expv[uu];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) bv[slot][uu][ct] = expv[2 + ct];
#else
      av[slot][uu] = wb[e4 * 4 * W15T + 4 * (u + uu) * NARROW];
#pragma unroll
      for (int ct = 0; ct < NC; ++ct) bv[slot][uu][ct] = ab[4 * (u + uu) * LDA_ + e4 * 4 * DIL_ + ct * 16];
#endif
    }
  };
  const bool last_q = kg != 3;                           // tap 15 (groups 15..19 of quarter 3) does not exist: wave-uniform
  auto live_ = [&](int g) { return g < NG && (g < 15 || last_q); };
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i) fetch(i);
  nsc_static_for<NG>([&](auto g_c) {
    constexpr int g = decltype(g_c)::value;
    if (live_(g)) {
      if (live_(g + DEPTH - 1)) fetch(g + DEPTH - 1);
      hook(g_c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int uu = 0; uu < 2; ++uu)
#pragma unroll
        for (int ct = 0; ct < NC; ++ct) acc[ct] = mfma4(av[g % DEPTH][uu], bv[g % DEPTH][uu][ct], acc[ct]);
      __builtin_amdgcn_sched_barrier(0);
    }
  });
}
// channels 16..19 PACKED as in d9_packed.  Dilation 1: row (s,i), column n -> dh[16+i][4n + s] = sum_{m,c'} A da[c'][4n + m],
// m = s + tap in [0, 18), A = w15s[m - s][c'][16+i] where 0 <= m - s < 15, else 0.
// Dilation 2 keeps the two time parities apart (a tap moves time by 2, so even outputs read even inputs only): column
// n = (parity n >> 3, n' = n & 7), row (s,i) -> dh[16+i][8n' + 2s + parity] = sum_{m,c'} A da[c'][8n' + parity + 2m] with the
// SAME A as dilation 1.  (Round 2 walked time with stride 4 at both dilations, m = s + 2 tap in [0, 32): for every m half of
// the shifts s had the wrong parity, so half the rows of every A fragment were zeros - 80 MFMAs per K-quarter instead of 50.)
// This wave owns the m = kg (mod 4).  One column tile covers the 64 output steps.  Two accumulators (even / odd
// k-steps, summed at the end) so that consecutive MFMAs are independent; half-groups of five are pipelined.
// d15_packed_col(kq, l15) = the time step of accumulator row group kq (= s), column l15
template <int DIL_>
__device__ __forceinline__ int d15_packed_col(int kq, int l15) {
  return DIL_ == 1 ? 4 * l15 + kq : 8 * (l15 & 7) + 2 * kq + (l15 >> 3);
}
template <int DIL_, int LDA_, int HALF>
__device__ __forceinline__ void d15_packed(const float* w15s, const float* da, int kg, int kq, int l15, f32x4& acc) {
  constexpr int NM = 4 + 14, NQM = (NM + 3) / 4, NH = 2 * NQM, H0 = HALF * (NH / 2), H1 = H0 + NH / 2;
  constexpr int QS = 4 * DIL_;                           // columns per step of m by 4
  const int sft = l15 >> 2, ich = l15 & 3;
  const float* bb = da + kq * LDA_ + (DIL_ == 1 ? 4 * l15 + kg : 8 * (l15 & 7) + (l15 >> 3) + 2 * kg);
  static_assert(8 * 7 + 1 + 2 * (NM - 1) < 64 + 14 * 2, "dilation 2: the last column read is inside the da window");
  constexpr int NS = NSC_D15_NS;                         // operands NS - 1 half-groups ahead (see d9_packed)
  float av[NS][5], bv[NS][5];
  bool okv[NS];
#if defined(NSC_EXP) && (NSC_EXP & 8)
  float expv[2] = {w15s[l15], bb[0]};
#endif
  auto fetch = [&](int hgrp, int slot) {
    const int q = hgrp >> 1, u0 = 5 * (hgrp & 1);
    const int tap = kg + 4 * q - sft;
    const bool ok = (unsigned)tap < (unsigned)K15;
    okv[slot] = ok;
    const float* ap = w15s + (ok ? tap : 0) * W15T + kq * NARROW + 16 + ich;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
#if defined(NSC_EXP) && (NSC_EXP & 8)
      av[slot][u] = expv[0];
      bv[slot][u] = expv[1];
#else
      av[slot][u] = ap[4 * (u0 + u) * NARROW];
      bv[slot][u] = bb[4 * (u0 + u) * LDA_ + QS * q];
#endif
    }
  };
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
  // only the LAST q can fall off the end (NM is not a multiple of 4 at DIL 1): live(q) is true for q < NQM - 1
  const bool last_live = kg + 4 * (NQM - 1) < NM;        // wave-uniform
  auto live_ = [&](int hgrp) { return hgrp < H1 && ((hgrp >> 1) < NQM - 1 || last_live); };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (live_(H0 + i)) fetch(H0 + i, (H0 + i) % NS);
#pragma unroll
  for (int hgrp = H0; hgrp < H1; ++hgrp) {
    if (live_(hgrp)) {
      if (live_(hgrp + NS - 1)) fetch(hgrp + NS - 1, (hgrp + NS - 1) % NS);
      __builtin_amdgcn_sched_barrier(0);
      const bool ok = okv[hgrp % NS];
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const float a = ok ? av[hgrp % NS][u] : 0.f;
        if ((u + hgrp) & 1) acc2 = mfma4(a, bv[hgrp % NS][u], acc2);
        else acc = mfma4(a, bv[hgrp % NS][u], acc);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] += acc2[r];
}

// -----------------------------------------------------------------------------------------------------
// v2 of the fused data-path backward: PERSISTENT and weight-stationary (same idea as gated_block_fwd2_kernel).
// One workgroup per CU walks (frame, 64-step tile) pairs; weights are fetched once per workgroup.  The data
// gradients have FEW output rows (20 -> two 16-row MFMA tiles) and LONG reductions (k9: 9*C, k15: 15*40), so the
// REDUCTIONS are split over the waves: wave w = (row tile w>>2, K-quarter w&3) owns k-steps {kg, kg+4, ...} of the
// k9 gradient (57 weight fragments, register-resident) and taps {kg, kg+4, kg+8, kg+12} of the k15 gradient (weights
// in LDS as [tap][40][20], shared by all waves), for ALL column tiles.  The four partial sums per output meet in LDS:
// plain ds_write of each wave's accumulators to a private slice, summed by the elementwise phase that follows
// (LDS float atomics measured ~2.4x slower than the MFMA work they were reducing).  Inner loops are {ds_read, v_mfma}.
// The next tile's dy / lin / tanh / h are prefetched into registers (raw buffer loads; out-of-frame columns come back
// as 0 from the bounds check) while this tile computes.
// -----------------------------------------------------------------------------------------------------
// CIN1: the block's INPUT has one channel (first block of a decoder stage): dy / dg / dh are as usual, but the 1x1 data
// gradient is a 20-term dot product per step and the residual branch sums dy over its C channels (the forward broadcast
// x over them); dx is one row and its producer is the quantizer (no activation gradient).
// ROLE: the kernel body is instantiated twice and dispatched on the (wave-uniform) wave index.  Waves w and w + 4 share a
// SIMD and a K-quarter (kg = w & 3) of both data gradients; ROLE = w >> 2 says which HALF of that quarter's work a wave does:
// two of the four dense column tiles (channels 0..15, k9 fragments in registers) AND half of the steps of the packed tile
// (channels 16..19, operands from LDS).  Until round 3 the dense tiles ran on waves 0-3 and the packed ones on waves 4-7: a
// wave streaming MFMAs from registers is never held up by its SIMD partner (tools/mfma_valu_coexec.hip), so the packed wave
// got the matrix pipe only when the dense one had finished and then ran alone, latency-bound (per-wave stamps: 9.8 k cycles
// dense, 13.4 k packed, of which the SIMD idled ~3 k; k15 gradient at dil 2: 7.8 k / 12.0 k).  With identical streams on both
// waves of a SIMD each fills the other's LDS waits and both end together.
// PAIRED: second block of a pair launch (nsc_pair_publish / nsc_pair_wait): its dy is what the first block's body wrote as dx
// FIRST: first block of a pair launch: dx leaves through write-through stores (see gated_block_fwd2_body)
template <int RT9, int NK9, int DIL, bool CIN1, int ROLE, bool PAIRED = false, bool FIRST = false>
__device__ __forceinline__ void gated_block_dgrad2_role(const BlockDgradArgs& a, int ntiles, int tpf, int skip, int* flags = nullptr, int* timeouts = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64, Hh = 7 * DIL, W_a = TT + 2 * Hh, W_dy = W_a + 8, NCTA = (W_a + 15) / 16, CR = 4 * NK9;
  // LDY / LDA == 14 (mod 32): the two channel rows a 32-lane group reads sit 14 banks apart.  The packed tiles walk time
  // with stride 4 (banks {0,4,..,28}): an offset that is not a multiple of 4 keeps the second row off the first row's
  // banks (at 112 == 16 they were the same banks: 4-way); the 16-consecutive-column reads of rows 0..15 overlap on 2 banks.
  constexpr int LDY = 110, LDA = 110, LDN = 80, WA16 = NCTA * 16;
  constexpr int PSW = WA16 + 4, PST = TT + 4;      // row strides of the partial-sum slices: == 4 (mod 8), so the four
                                                   // D-fragment rows a wave stores (4 rows apart) hit four bank groups
  constexpr int NQ = (CR + 7) / 8;                 // dy rows per wave
  constexpr int NJ9 = (NK9 + 3) / 4;               // channel groups (of 4) per K-quarter of the k9 gradient
  constexpr int PART1 = 4 * 16 * PSW;              // offset of the row-tile-1 partial sums (rows 16..19 only)
  constexpr int LDD = 68;                          // dxs row stride: == 4 (mod 32), D-fragment stores conflict-free
  constexpr int PARTSZ = 4 * NARROW * PSW > CR * LDD ? 4 * NARROW * PSW : CR * LDD;
  static_assert(WA16 + 8 <= LDY && 63 + 15 * DIL < LDA, "tile widths");
  // The staged tiles sit DLT columns to the right of their logical position, so that physical column 0 of a row is a time
  // that is a multiple of 4: the tiles then arrive as ALIGNED 16-byte loads that lie wholly inside or wholly outside a frame
  // row (T a multiple of 4; else per-element masks), zeros for the outside come from the buffer bounds check.
  constexpr int DLT = (4 - (Hh & 3)) & 3;          // Hh + DLT == 0 (mod 4)
  float* dys = sm + DLT;                           // [CR][LDY]       j  <-> t0 - Hh - 4 + j   (pad rows zero)
  float* lin = sm + CR * LDY + DLT;                // [20][LDA]       ja <-> t0 - Hh + ja      -> dlin  (rows 0..19 of da)
  static_assert(WA16 + 8 + DLT <= LDY && 64 + 15 * DIL + DLT <= LDA && (LDY % 2 == 0) && (LDA % 2 == 0) && ((CR * LDY) % 2 == 0),
                "shifted tiles fit their rows; 8-byte aligned row bases");
  float* th = lin + NARROW * LDA;                  // [20][LDA]                                 -> dgate (rows 20..39 of da)
  float* dhs = sm + CR * LDY + 2 * NARROW * LDA;   // [20][LDN]       tt <-> t0 + tt : dz1   (16-byte aligned, like all below)
  static_assert((CR * LDY + 2 * NARROW * LDA) % 4 == 0 && LDN % 4 == 0 && PST % 4 == 0 && PARTSZ % 4 == 0, "float4 LDS rows");
  float* part = dhs + NARROW * LDN;                // [4][16][WA16] + [4][4][WA16]  partial sums of the K-quarters;
  float* dxs = part;                               //   later [CR][LDD]: dx before act'(x), for the row-wise copy-out
  float* w15s = part + PARTSZ;                     // [15][40][20]   wt_l | wt_r concatenated along the reduced channel
  float* w9ps = w15s + K15 * W15T;                 // [9][C][4] (tap stride w9t)  k9 gradient weights of channels 16..19
  const int C = a.C, T = a.T;
  const int w9t = C * 4 + (((C * 4) & 15) == 8 ? 0 : 8);   // == 8 or 24 (mod 32): see W15T
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  constexpr int hf = ROLE;                         // which half of a K-quarter's work this wave does (see the kernel)
  const int kg = wave & 3;                         // this wave's K-quarter: waves w and w + 4 share it (and a SIMD)

  // ---- buffer descriptors of the tensors this kernel streams (out-of-frame pieces are sent past num_records: zeros) ----
  const __amdgpu_buffer_rsrc_t sdy =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)((long)a.B * C * T * 4), 0x00020000);
  const unsigned nbN = (unsigned)((long)a.B * NARROW * T * 4);
  const __amdgpu_buffer_rsrc_t slin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.lin), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sth = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.th), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sxx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * C * T * 4), 0x00020000);
  // da / dgate: two views of one [B][da_rows][T] tensor (dgate = da + 20 T) or two [B][20][T] tensors: the range check of either
  // resource covers what is addressable from its base
  const unsigned nbDA = (unsigned)(((long)a.B * a.da_rows - (a.da_rows == NARROW ? 0 : NARROW)) * T * 4);
  const __amdgpu_buffer_rsrc_t sdxo = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, FIRST ? (unsigned)((long)a.B * C * T * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t sdlin = __builtin_amdgcn_make_buffer_rsrc(a.da, 0, nbDA, 0x00020000);
  const __amdgpu_buffer_rsrc_t sdgate = __builtin_amdgcn_make_buffer_rsrc(a.dgate, 0, nbDA, 0x00020000);
  // ---- prefetch of the next tile, 16 bytes per lane: lane = (row half lane >> 5, float4 index lane & 31); a wave instruction
  // covers two rows.  dy: wave w rows 2 w + half + 16 q; lin / tanh: rows 2 w + half (+ 16: waves 0, 1); h: float4 tid of the
  // [20][64] tile.  (Round 2 moved these as dword loads + ds_write_b32, one float per lane and instruction; 16 bytes per lane took
  // ~2 k cycles out of the staging and copy-out phases of a tile.)
  constexpr int NQY = (CR + 15) / 16, NY4 = (W_dy + DLT + 3) / 4, NA4 = (W_a + DLT + 3) / 4;
  static_assert(NY4 <= 32 && NA4 <= 32, "a row's window fits 32 float4 lanes");
  f32x4 pfy[NQY], pfl[2], pft[2], pfh;
  const int pi4 = lane & 31, phalf = lane >> 5;
  const bool tvec = (T & 3) == 0;                  // rows are 16-byte aligned: float4 pieces never straddle a row end
  auto bld4 = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NSC_AUX_STREAM));
  };
  auto bld = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  const int OOB = 0x7ffffff0;
  // per-lane byte offset of (time ty0 + 4 pi4) inside a row, or out of range when the piece lies outside [0, T) (T % 4 == 0)
  // or at least starts outside it (general T: the pieces that straddle the row end are masked per element when staged)
  // The 12 loads of a prefetch are NOT issued together: a VMEM instruction takes the issuing wave ~100 cycles under load, and
  // issued in one piece after a barrier all 8 waves pay that at the same time with the matrix pipe idle (4 loads: 500 cycles of
  // every tile).  pf_setup computes the tile's offsets; the loads go out one at a time from hooks inside the two long MFMA
  // loops, where the issuing wave's SIMD partner keeps the matrix pipe busy (the k15-gradient phase got 1.2 k cycles shorter).
  // The row part of an offset must be wave-uniform where it sits in the scalar operand: see pf_vy.
  // (The same interleaving made the weight-gradient kernels - ~50 DWORD loads per tile with scalar address arithmetic each -
  // 20 % slower: conv wgrad 100->100 k9 s2 went from 110 to 133 us.  It pays for a handful of 16-byte loads, not for those.)
  int pf_vy = 0, pf_va = 0, pf_vh = 0, pf_b = 0;
  auto pf_setup = [&](int tile, bool steady) {
    const int tl = __builtin_amdgcn_readfirstlane(tile < ntiles ? tile : 0);
    const int b = tl / tpf, t0 = (tl - b * tpf) * TT;
    pf_b = b;
    const int ty = t0 - Hh - 4 - DLT + 4 * pi4;
    // a steady tile reads its dy window from column Hh + 4 on (the residual path; the k9 gradient from 2 Hh): the rest is not fetched
    const bool need = pi4 < NY4 && ty >= 0 && ty < T && !(steady && 4 * pi4 + 3 < Hh + 4 + DLT);
    // (the per-lane part of the row - the half-wave - sits in the VECTOR offset: a non-uniform scalar offset makes the compiler
    // wrap every load in a readfirstlane "waterfall" loop)
    pf_vy = need ? (phalf * T + ty) * 4 : OOB;
    const int ta = t0 - Hh - DLT + 4 * pi4;
    const bool needa = pi4 < NA4 && ta >= 0 && ta < T && !(steady && 4 * pi4 + 3 < 2 * Hh + DLT);
    pf_va = needa ? (phalf * T + ta) * 4 : OOB;
    const int c = tid >> 4, t = t0 + 4 * (tid & 15);          // h: float4 tid of the [20][64] dz1 tile (tid < 320)
    pf_vh = (c < NARROW && t < T) ? ((lane >> 4) * T + t) * 4 : OOB;
  };
  // (rows past the last channel would read the next frame's rows - the staging step zeroes them, but they are HBM traffic:
  // 112 rows fetched for 100.  Only the last q of a wave can reach them: those lanes' offset goes out of range.)
  auto pf_dy1 = [&](int q) {
    const int vo = (q == NQY - 1 && 2 * wave + phalf + 16 * q >= C) ? OOB : pf_vy;
    pfy[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sdy, vo, (pf_b * C + 2 * wave + 16 * q) * T * 4,
                                                                             PAIRED ? NSC_AUX_COHERENT : NSC_AUX_STREAM));
  };
  auto pf_a1 = [&](int i) {                                    // i = 0..4: lin rows, tanh rows, lin rows + 16, tanh rows + 16, h
    const int so = (pf_b * NARROW + 2 * wave + 16 * (i >> 1)) * T * 4;
    const int va1 = 2 * wave + phalf + 16 >= NARROW ? OOB : pf_va;       // rows 20..31 of the second pass do not exist
    if (i == 0) pfl[0] = bld4(slin, pf_va, so);
    else if (i == 1) pft[0] = bld4(sth, pf_va, so);
    else if (i == 2) pfl[1] = bld4(slin, va1, so);
    else if (i == 3) pft[1] = bld4(sth, va1, so);
    else pfh = bld4(sh, pf_vh, (pf_b * NARROW + 4 * wave) * T * 4);
  };
  auto prefetch_dy = [&]() {
#pragma unroll
    for (int q = 0; q < NQY; ++q) pf_dy1(q);
  };
  auto prefetch_a = [&]() {
#pragma unroll
    for (int i = 0; i < 5; ++i) pf_a1(i);
  };
  // CHAIN: a workgroup walks CONSECUTIVE tiles [first, last).  Consecutive tiles of one frame overlap in the 2*Hh columns of
  // da (= dlin | dgate) that the k15 gradient reads on either side: from the second tile of a chain on, those columns are
  // carried over in LDS and the k9 gradient + GLU run on the TT new columns only (4 column tiles instead of 6: the halo was
  // 1.44x of the k9 gradient's MFMAs).  A tile is "fresh" (full width) at the start of a chain and at the start of a frame.
  const int first = (int)((long)blockIdx.x * ntiles / gridDim.x), last = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  NSC_STAMP(0);
  pf_setup(first, false);
#if !(defined(NSC_EXP) && (NSC_EXP & 1))
  if (!PAIRED) prefetch_dy();            // (a paired body: after the neighbours' flags, below)
  prefetch_a();
#endif
  static_assert(NK9 == 4 * (NJ9 - 1) + 1, "left-over channel group is shared out by tap");
  float w9r[K9][NJ9 - 1], w9x[3];
  const int rt1 = RT9 == 7 ? (wave < 7 ? wave : 6) : (wave & 3);
  const int cb1 = RT9 == 7 ? 0 : (wave >> 2) * 32;
  constexpr int NC1 = RT9 == 7 ? 4 : 2;
  float w1r[5];
  if (a.img) {
    // FAST prologue from the engine's kernel-ready image (see gated_block_fwd2_kernel).  Round 4: only what the FIRST MFMA phase
    // needs is waited for here - the k9^T fragments, the packed k9 table (w9ps, through registers) and the first tile.  The k15
    // table (w15s, 48 KB: a quarter of the image) is not read before the first k15-gradient phase: it goes straight into LDS by
    // LDS-DMA (global_load_lds_dwordx4, no registers) issued AFTER the wait below and lands under the first k9-gradient phase.
    // The fragments of waves w and w + 4 (same K-quarter) are ONE copy in the image: both waves fetch the same lines (the
    // second fetch is an L1 hit or merges with the first) - the per-wave copies were 131 KB of L2 -> CU traffic per workgroup,
    // x 256 workgroups bursting at once, the prologue is bound by exactly that.
    const f32x4* img4 = reinterpret_cast<const f32x4*>(a.img);
    constexpr int N15_4 = K15 * W15T / 4;
    const int n9_4 = K9 * w9t / 4;
    constexpr int NE9 = (K9 * (4 * 4 * NK9 + 8) / 4 + 511) / 512;
    constexpr int NFS = K9 * (NJ9 - 1) + 3, NF4S = (NFS + 3) / 4, NF4W = 2;
    f32x4 t9[NE9], fr[NF4S], fw[NF4W];
#if defined(NSC_EXP) && (NSC_EXP & 2)
#pragma unroll
    for (int i = 0; i < NE9; ++i) t9[i] = (f32x4){0.f, 0.f, 0.f, (float)tid};
#pragma unroll
    for (int g = 0; g < NF4S; ++g) fr[g] = (f32x4){0.f, 0.f, (float)g, (float)tid};
#pragma unroll
    for (int g = 0; g < NF4W; ++g) fw[g] = (f32x4){0.f, 0.f, (float)g, (float)tid};
#else
#pragma unroll
    for (int i = 0; i < NE9; ++i) t9[i] = img4[N15_4 + min(tid + 512 * i, n9_4 - 1)];
    const f32x4* fps = img4 + N15_4 + n9_4 + kg * NF4S * 64 + lane;
#pragma unroll
    for (int g = 0; g < NF4S; ++g) fr[g] = fps[g * 64];
    const f32x4* fpw = img4 + N15_4 + n9_4 + 4 * NF4S * 64 + wave * NF4W * 64 + lane;
#pragma unroll
    for (int g = 0; g < NF4W; ++g) fw[g] = fpw[g * 64];
#endif
#pragma unroll
    for (int i = 0; i < NE9; ++i)
      if (tid + 512 * i < n9_4) reinterpret_cast<f32x4*>(w9ps)[tid + 512 * i] = t9[i];
#define NSC_FR(f) fr[(f) / 4][(f) % 4]
#pragma unroll
    for (int tp = 0; tp < K9; ++tp)
#pragma unroll
      for (int j = 0; j < NJ9 - 1; ++j) w9r[tp][j] = NSC_FR(tp * (NJ9 - 1) + j);
#pragma unroll
    for (int i = 0; i < 3; ++i) w9x[i] = NSC_FR(K9 * (NJ9 - 1) + i);
#undef NSC_FR
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) w1r[s5] = CIN1 ? 0.f : fw[s5 / 4][s5 % 4];
  } else {
  // ---- once per workgroup: weights -> registers / LDS (clamped indices: pad k-rows meet zero rows of the staged
  // tiles; only k-steps / taps past the end need a real zero) ----
  // (all loads of a table are issued before its first LDS store: a plain load->store loop serialises ~24 L2 round
  // trips per lane, which was ~9 us of every launch)
  {
    constexpr int N15 = K15 * 2 * NARROW * NARROW, NE = (N15 + 511) / 512;
    float tmp[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = min(tid + 512 * i, N15 - 1);
      const int tp = e / (2 * NARROW * NARROW), r = e - tp * 2 * NARROW * NARROW;
      const int cp = r / NARROW, ci = r - cp * NARROW;
      const float* src = cp < NARROW ? a.wtl : a.wtr;
      tmp[i] = src[(tp * NARROW + (cp < NARROW ? cp : cp - NARROW)) * NARROW + ci];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 512 * i;
      const int tp = e / (2 * NARROW * NARROW);
      if (e < N15) w15s[e + 8 * tp] = tmp[i];
    }
  }
  // k9 gradient: K-quarter kg owns the channel groups cq = kg + 4j of every tap, so the B-fragment address of step
  // (tap', j) is lane base + the compile-time offset (16 j LDY + tap'): no per-step address registers.
  {
    constexpr int NE9 = (K9 * 4 * 4 * NK9 + 511) / 512;
    const int n9 = K9 * C * 4;
    float tmp[NE9];
#pragma unroll
    for (int i = 0; i < NE9; ++i) {
      const int e = min(tid + 512 * i, n9 - 1);
      tmp[i] = a.wt9[(long)(e >> 2) * NARROW + 16 + (e & 3)];
    }
#pragma unroll
    for (int i = 0; i < NE9; ++i) {
      const int e = tid + 512 * i;
      if (e < n9) w9ps[(e / (4 * C)) * w9t + (e % (4 * C))] = tmp[i];
    }
  }
  // NK9 = 4 (NJ9-1) + 1 for both shapes: the one left-over channel group (cq = NK9-1) is shared out by TAP (quarter kg takes
  // taps kg, kg+4, kg+8 < 9), so the quarters carry 57 | 56 | 56 | 56 k-steps instead of 63 | 54 | 54 | 54.
#pragma unroll
  for (int tp = 0; tp < K9; ++tp)
#pragma unroll
    for (int j = 0; j < NJ9 - 1; ++j)
      w9r[tp][j] = a.wt9[((long)tp * C + min(4 * (kg + 4 * j) + kq, C - 1)) * NARROW + l15];   // rows = channels 0..15
#pragma unroll
  for (int i = 0; i < 3; ++i)
    w9x[i] = a.wt9[((long)min(kg + 4 * i, K9 - 1) * C + min(4 * (NK9 - 1) + kq, C - 1)) * NARROW + l15];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5) w1r[s5] = CIN1 ? 0.f : a.wt1[(s5 * 4 + kq) * C + min(rt1 * 16 + l15, C - 1)];
  }

  // CIN1: wt1 is [20] (the flipped / transposed [1,1,20] kernel); lane tt of wave 0 needs all of it
  float w1c[CIN1 ? NARROW : 1];
  if (CIN1) {
    // (with an image: the 20 taps follow the fragment region)
    const float* w1p = a.img ? a.img + (K15 * W15T + K9 * w9t) + 64 * 4 * (4 * ((K9 * (NJ9 - 1) + 3 + 3) / 4) + 8 * 2) : a.wt1;
#pragma unroll
    for (int c = 0; c < NARROW; ++c) w1c[c] = w1p[c];
  }


  if (PAIRED) {
    nsc_pair_wait(flags, timeouts);                // the first block's dx around this workgroup's tiles is in memory
    prefetch_dy();
  }
  nsc_wait_vmem();   // weights (and the first tile) are in: no vmcnt guards on register operands inside the loop
  // the k15 table by LDS-DMA (image path): 16 bytes per lane, 1 KiB per wave instruction, destination m0 + lane * 16.  Hidden in
  // inline asm: hipcc would otherwise wait vmcnt(0) for it at the next use of any ordinary load's result and at every barrier.
  // Issued after the wait above (it is the YOUNGEST vector-memory operation when the tile loop starts) and waited for by hand at
  // the end of the first tile's k9-gradient phase.
  const bool dma15 = a.img != nullptr;
  if (dma15) {
#if !(defined(NSC_EXP) && (NSC_EXP & 2))
    constexpr int N15_4 = K15 * W15T / 4, NG15 = (N15_4 + 511) / 512;
    const unsigned lds15 = (unsigned)(unsigned long long)w15s;
#pragma unroll
    for (int i = 0; i < NG15; ++i) {
      if (tid + 512 * i < N15_4) {
        const f32x4* gp = reinterpret_cast<const f32x4*>(a.img) + tid + 512 * i;
        const unsigned m0v = lds15 + (unsigned)((i * 8 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp), "s"(m0v) : "memory", "m0");
      }
    }
#endif
  }
  NSC_STAMP(1);
  for (int tile = first; tile < last; ++tile) {
    const int b = tile / tpf, t0 = (tile - b * tpf) * TT;
    const bool fresh = tile == first || t0 == 0;                                   // workgroup-uniform
    const bool next_steady = tile + 1 < last && (tile + 1) - ((tile + 1) / tpf) * tpf != 0;
    NSC_STAMP(2);
    // ---- phase 0: prefetched tiles -> LDS (8-byte stores: the row strides are even, not multiples of 4) ----
    {
      // (indices from an opaque copy of the lane id: left visible, the compiler hoists the ~15 LDS addresses of this phase out
      // of the tile loop and holds them across the MFMA phases - the registers those need for operand double-buffering)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int pi4 = lane_o & 31, phalf = lane_o >> 5;
      float* dyp = dys - DLT;                      // physical row bases
      float* lip = lin - DLT;
      float* thp = th - DLT;
      if (pi4 < LDY / 4) {
        const int ty = t0 - Hh - 4 - DLT + 4 * pi4;
#pragma unroll
        for (int q = 0; q < NQY; ++q) {
          const int r = 2 * wave + phalf + 16 * q;
          // (only the last q can reach rows past CR; rows past C - what the loads fetched there belongs to the next frame - only
          // where 16 q + 15 >= C: a scalar test first, most q never run the per-lane select)
          if (q < NQY - 1 || r < CR) {
            f32x4 v = pfy[q];
            if (16 * q + 15 >= C) {
              if (r >= C) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (!tvec) {                                  // a row ends inside this piece: the rest came from the next row
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (unsigned)(ty + e) < (unsigned)T ? v[e] : 0.f;
            }
            float2* dst = reinterpret_cast<float2*>(dyp + r * LDY + 4 * pi4);
            dst[0] = make_float2(v[0], v[1]);
            dst[1] = make_float2(v[2], v[3]);
          }
        }
      }
      if (pi4 < NA4 + 1 && 4 * pi4 + 3 < LDA) {
        const int ta = t0 - Hh - DLT + 4 * pi4;
        constexpr int BND = 2 * Hh + DLT;                 // first physical column of a steady tile's new dlin / dgate
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = 2 * wave + phalf + 16 * q;
          if (r < NARROW) {
            f32x4 vl = pfl[q], vt = pft[q];
            if (!tvec) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const bool in = (unsigned)(ta + e) < (unsigned)T;
                vl[e] = in ? vl[e] : 0.f;
                vt[e] = in ? vt[e] : 0.f;
              }
            }
            // steady tile: physical columns [DLT, BND) hold the carried dlin / dgate and must stay
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
              const int p0 = 4 * pi4 + 2 * hp;
              float* dl_ = lip + r * LDA + p0;
              float* dt_ = thp + r * LDA + p0;
              if (fresh || p0 >= BND) {
                *reinterpret_cast<float2*>(dl_) = make_float2(vl[2 * hp], vl[2 * hp + 1]);
                *reinterpret_cast<float2*>(dt_) = make_float2(vt[2 * hp], vt[2 * hp + 1]);
              } else if (p0 + 1 >= BND) {                 // (odd boundary: dil 1)
                dl_[1] = vl[2 * hp + 1];
                dt_[1] = vt[2 * hp + 1];
              }
            }
          }
        }
      }
    }
    // lrelu'(h) of this lane's four dz1 elements, as sign bits (one register across the two long MFMA phases)
    const unsigned hpos = (pfh[0] > 0.f ? 1u : 0u) | (pfh[1] > 0.f ? 2u : 0u) | (pfh[2] > 0.f ? 4u : 0u) | (pfh[3] > 0.f ? 8u : 0u);
    NSC_STAMP(3);
    nsc_lds_barrier();
    NSC_STAMP(4);
    // the next tile's dy goes out during the k9 gradient, lin / tanh / h during the k15 gradient, one load at a time (see
    // pf_setup).  Probe runs that skip phases issue them in one piece.
    pf_setup(tile + 1 < last ? tile + 1 : tile, next_steady);
    const bool spread = skip == 0;
    if (!spread && !(skip & 8)) prefetch_dy();
    auto hook9 = [&](auto st_c) {
      constexpr int st = decltype(st_c)::value, NST = K9 * (NJ9 - 1), EV = NST / (NQY + 1);
      if constexpr (st % EV == 0 && st > 0 && st / EV <= NQY) {
        if (spread) pf_dy1(st / EV - 1);
      }
    };
    auto hook15 = [&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      if constexpr (g % 3 == 1 && g < 15) {
        if (spread) pf_a1(g / 3);
      }
    };

    // ---- D9: dg[ci][ja] = sum_{tap', o} wt9[tap'][o][ci] * dy[o][ja + tap'], K split in quarters kg.
    // Waves w and w + 4 share K-quarter kg = w & 3 (and a SIMD) and run the same stream: two of the four dense column tiles of
    // channels 0..15 and half of the packed-tile steps of channels 16..19 each (gated_block_dgrad2_role's ROLE).
    // Software-pipelined loops with scheduling fences: left alone the scheduler hoists ~30 ds_read2 (60 registers) ahead
    // of the MFMAs, which spills - and a scratch reload waits on vmcnt IN ORDER, i.e. on the next tile's whole prefetch.
    if (!(skip & 1)) {
      // dense rows 0..15: this wave's half of the column tiles (steady: 2 of the 4 new tiles; fresh: NA | NCTA - NA of all)
      // (the row stride 110 == 14 (mod 32) makes lanes kq = 0 / 1 of this read overlap in two banks - a 2-way conflict on every
      // dense operand read, SQ_LDS_BANK_CONFLICT ~ 5 k cycles per tile; a build with conflict-free rows ran no faster: the LDS
      // array is < 50 % busy and the loops wait on MFMA issue, not on it.  profiles/r03d_dgrad_experiments.txt)
      const float* yb = dys + (4 * kg + kq) * LDY + l15;
      const float* yx = dys + (4 * (NK9 - 1) + kq) * LDY + l15 + kg;
      float* pp0 = part + (kg * 16 + kq * 4) * PSW + l15;
      // channels 16..19, packed: this wave's half of the steps; the second half's partial sums go to the (idle) dz1 tile
      float* ppk = (hf == 0 ? part + PART1 : dhs) + kg * 4 * PSW;
      static_assert(16 * PSW <= NARROW * LDN, "second set of packed partial sums lives in the dz1 tile");
      if (!fresh) {
        constexpr int J0 = 2 * Hh, JB = J0 + 32 * hf;
        {
          f32x4 acc[2];
          acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
          d9_rows0<2, NJ9 - 1, LDY>(w9r, w9x, yb + JB, yx + JB, kg, acc, hook9);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pp0[reg * PSW + JB + ct * 16] = acc[ct][reg];
        }
        f32x4 pk[2];
        pk[0] = pk[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        d9_packed<NJ9 - 1, NK9, LDY, 1, hf>(w9ps, C, w9t, dys + J0, kg, kq, l15, pk);
        const int col = J0 + 4 * l15 + kq;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ppk[reg * PSW + col] = pk[0][reg] + pk[1][reg];
      } else {
#if defined(NSC_EXP) && (NSC_EXP & 32)
        // timing experiment (wrong values): what an EDGE-started chain would save - a fresh tile at t = 0 needs no dg left of the
        // frame, i.e. one dense column tile less at dilation 2 (78 of 92 columns: 5 tiles instead of 6; dilation 1: 71 of 78, still 5)
        constexpr int NCTE = DIL == 2 ? NCTA - 1 : NCTA;
        constexpr int NA = (NCTE + 1) / 2, NMINE = hf == 0 ? NA : NCTE - NA, CB = hf == 0 ? 0 : NA * 16;
#else
        constexpr int NA = (NCTA + 1) / 2, NMINE = hf == 0 ? NA : NCTA - NA, CB = hf == 0 ? 0 : NA * 16;
#endif
        {
          f32x4 acc[NMINE];
#pragma unroll
          for (int ct = 0; ct < NMINE; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
          d9_rows0<NMINE, NJ9 - 1, LDY>(w9r, w9x, yb + CB, yx + CB, kg, acc, hook9);
#pragma unroll
          for (int ct = 0; ct < NMINE; ++ct)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) pp0[reg * PSW + CB + ct * 16] = acc[ct][reg];
        }
        f32x4 pk[2];
        pk[0] = pk[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        d9_packed<NJ9 - 1, NK9, LDY, 2, hf>(w9ps, C, w9t, dys, kg, kq, l15, pk);
        // row (s = kq, i = reg), column n = l15  ->  dg[16 + reg][64 ctp + 4 l15 + kq]
#pragma unroll
        for (int ctp = 0; ctp < 2; ++ctp) {
          const int col = 64 * ctp + 4 * l15 + kq;
          if (col < WA16) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ppk[reg * PSW + col] = pk[ctp][reg];
          }
        }
      }
    }
    NSC_STAMP(5);
    if (dma15 && tile == first) {
      // the k15 table's LDS-DMA has had the whole k9-gradient phase to land.
      // (vmcnt(0), not a count of the younger prefetch loads - ADVICE r4: a counted wait breaks silently when a load or store is added
      // between the DMA and this point; first tile of a launch only, and the dy prefetch it also waits for went out a phase ago)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    nsc_lds_barrier();
    NSC_STAMP(6);

    // ---- GLU backward in place (lin/th are zero outside the frame, so dlin/dgate are too); da -> global ----
    // Columns: a fresh tile has all WA16 columns of partial sums, a steady one the TT new columns [2 Hh, W_a).  da leaves for
    // HBM for this tile's own range [Hh, Hh + TT) - and, when the next tile is steady, for the right halo too: those values
    // are final (the next tile would recompute exactly them) and the next tile then never touches its carried columns.
    {
      // (column count as a compile-time constant - the item -> (channel, column) split is a shift or a multiply, not a
      // division - and da / dgate leave through buffer stores with a scalar frame offset: this phase is VALU-issue bound)
      const int st_hi = next_steady ? W_a : Hh + TT;
      const int sda = b * a.da_rows * T * 4;
      int tid_g = tid;
      asm volatile("" : "+v"(tid_g));
      auto glu = [&](auto fresh_c) {
        constexpr bool FR = decltype(fresh_c)::value;
#if defined(NSC_EXP) && (NSC_EXP & 32)
        constexpr int j_lo = FR ? 0 : 2 * Hh, ncol = FR ? (DIL == 2 ? WA16 - 16 : WA16) : TT, st_lo = FR ? Hh : 2 * Hh;
#else
        constexpr int j_lo = FR ? 0 : 2 * Hh, ncol = FR ? WA16 : TT, st_lo = FR ? Hh : 2 * Hh;
#endif
        constexpr int NIT = (NARROW * ncol + 511) / 512;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int e = tid_g + 512 * it;
          if (NARROW * ncol % 512 == 0 || it + 1 < NIT || e < NARROW * ncol) {
            const int c = e / ncol, ja = j_lo + (e - c * ncol);
            const float* pp = c < 16 ? part + c * PSW + ja : part + PART1 + (c - 16) * PSW + ja;
            const int ps = c < 16 ? 16 * PSW : 4 * PSW;
            float gg = (pp[0] + pp[ps]) + (pp[2 * ps] + pp[3 * ps]);
            if (c >= 16) {                                     // the second halves of the packed K-quarters
              const float* pq = dhs + (c - 16) * PSW + ja;
              gg += (pq[0] + pq[ps]) + (pq[2 * ps] + pq[3 * ps]);
            }
            float* pl = lin + c * LDA + ja;
            float* pt = th + c * LDA + ja;
            const float l = *pl, tg = *pt;
            const float dl_ = gg * tg, dgt = gg * l * (1.f - tg * tg);
            *pl = dl_;
            *pt = dgt;
            const int t = t0 - Hh + ja;
            const int vo = (ja >= st_lo && ja < st_hi && t < T) ? (c * T + t) * 4 : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, dl_), sdlin, vo, sda, NSC_AUX_LATE);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, dgt), sdgate, vo, sda, NSC_AUX_LATE);
          }
        }
      };
      if (fresh) glu(std::true_type{});
      else glu(std::false_type{});
    }
    NSC_STAMP(7);
    nsc_lds_barrier();
    NSC_STAMP(8);

    // ---- D15: dh[ci][tt] = sum wt_lr[tap'][c'][ci] * da[c'][tt + tap' d], taps split in quarters kg.
    // Same split as the k9 gradient: two dense column tiles and half of the packed-tile steps per wave.
    if (!(skip & 2)) {
      const float* ab = lin + kq * LDA + l15 + kg * DIL + 32 * hf;
      const float* wb = w15s + kg * W15T + kq * NARROW + l15;
      float* pp0 = part + (kg * 16 + kq * 4) * PST + l15 + 32 * hf;
      {
        f32x4 acc[2];
        acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        d15_rows0<2, DIL, LDA>(wb, ab, kg, acc, hook15);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) pp0[reg * PST + ct * 16] = acc[ct][reg];
      }
      f32x4 pk = {0.f, 0.f, 0.f, 0.f};
      d15_packed<DIL, LDA, hf>(w15s, lin, kg, kq, l15, pk);
      // row (s = kq, i = reg), column n = l15  ->  dh[16 + reg][time step of (kq, l15)]; eight partial sums per element
      // (K-quarter, half)
      static_assert((64 + 32) * PST <= PARTSZ, "partial sums of the k15 gradient");
      const int pcol = d15_packed_col<DIL>(kq, l15);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) part[4 * 16 * PST + ((hf * 4 + kg) * 4 + reg) * PST + pcol] = pk[reg];
    }
    NSC_STAMP(9);
    nsc_lds_barrier();
    NSC_STAMP(10);
    if (!spread && !(skip & 8)) prefetch_a();
    // x rows of the copy-out phase (for act'(x)): float4 (row, 4 q4) items of the [C][64] dx tile, row = tid / 16 + 32 q,
    // q4 = tid % 16; consumed two barriers later
    constexpr int NQX = (CR + 31) / 32;
    f32x4 xv[NQX];
    if (!CIN1 && a.in_act == NSC_ACT_LRELU) {
      const int tx = t0 + 4 * (tid & 15);
      const int vx = tx < T ? ((lane >> 4) * T + tx) * 4 : OOB;
#pragma unroll
      for (int q = 0; q < NQX; ++q)
        xv[q] = bld4(sxx, (q == NQX - 1 && 4 * wave + (lane >> 4) + 32 * q >= C) ? OOB : vx, (b * C + 4 * wave + 32 * q) * T * 4);
    }
    NSC_STAMP(18);
    if (next_steady) {
      // carry: the last 2 Hh columns of dlin / dgate (columns [TT, TT + 2 Hh) of this tile) are columns [0, 2 Hh) of the next
      // tile.  The k15 gradient is done with them (barrier above) and source / destination ranges are disjoint.
      if constexpr (DLT % 2 == 0) {                        // pairs of columns (8-byte aligned: the row bases and the shift are even)
        for (int e = tid; e < 2 * NARROW * Hh; e += 512) {
          const int r = e / Hh, cidx = 2 * (e - r * Hh);
          float* buf = r < NARROW ? lin + r * LDA : th + (r - NARROW) * LDA;
          *reinterpret_cast<float2*>(buf + cidx) = *reinterpret_cast<const float2*>(buf + cidx + TT);
        }
      } else {
        for (int e = tid; e < 2 * NARROW * 2 * Hh; e += 512) {
          const int r = e / (2 * Hh), cidx = e - r * (2 * Hh);
          float* buf = r < NARROW ? lin + r * LDA : th + (r - NARROW) * LDA;
          buf[cidx] = buf[cidx + TT];
        }
      }
    }

    // ---- dz1 = (sum of the partial dh) . lrelu'(h) -> LDS + global: one float4 (channel c, steps 4 q4 ..) per lane ----
    NSC_STAMP(19);
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));                // (as in the staging phase: addresses recomputed per tile, not held)
    if (tid_o < NARROW * 16) {
      const int c = tid_o >> 4, tt = 4 * (tid_o & 15);
      const int t = t0 + tt;
      const float* pp = c < 16 ? part + c * PST + tt : part + 4 * 16 * PST + (c - 16) * PST + tt;
      const int ps = c < 16 ? 16 * PST : 4 * PST;
      auto ld4 = [](const float* q_) { return *reinterpret_cast<const f32x4*>(q_); };
      f32x4 dh = (ld4(pp) + ld4(pp + ps)) + (ld4(pp + 2 * ps) + ld4(pp + 3 * ps));
      if (c >= 16) dh += (ld4(pp + 4 * ps) + ld4(pp + 5 * ps)) + (ld4(pp + 6 * ps) + ld4(pp + 7 * ps));   // second halves of the packed K-quarters
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = t + e < T ? dh[e] * ((hpos >> e) & 1u ? 1.f : NSC_LRELU_ALPHA) : 0.f;
      *reinterpret_cast<f32x4*>(dhs + c * LDN + tt) = v;
      float* gp = a.dz1 + ((long)b * NARROW + c) * T + t;
      if (tvec) {
        if (t < T) nsc_store4_late(gp, v);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (t + e < T) gp[e] = v[e];
      }
    }
    NSC_STAMP(11);
    nsc_lds_barrier();
    NSC_STAMP(12);

    if (CIN1) {
      // ---- D1, one input channel: dx[t] = sum_c w1[c] dz1[c][t] + sum_o dy[o][t].  Wave w sums the dy rows w, w+8, ...
      // (wave 0 adds the 20-term dot product) into dxs[w][.]; after the barrier wave 0 adds the eight partial rows.
      const int tt = lane;
      float s_ = 0.f;
      for (int o = wave; o < C; o += 8) s_ += dys[o * LDY + tt + Hh + 4];
      if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NARROW; ++c) s_ = fmaf(w1c[c], dhs[c * LDN + tt], s_);
      }
      dxs[wave * LDD + tt] = s_;
      NSC_STAMP(13);
      nsc_lds_barrier();
      NSC_STAMP(14);
      if (wave == 0 && t0 + tt < T) {
        float v = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) v += dxs[w8 * LDD + tt];
        a.dx[(long)b * T + t0 + tt] = v;
      }
    } else {
    // ---- D1: dx = (W1^T dz1 + dy) . act'(x); this wave's row tile, NC1 column tiles ----
      if ((RT9 != 7 || wave < 7) && !(skip & 4)) {
        f32x4 acc[NC1];
  #pragma unroll
        for (int c = 0; c < NC1; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* zb = dhs + kq * LDN + cb1 + l15;
  #pragma unroll
        for (int s5 = 0; s5 < 5; ++s5)
  #pragma unroll
          for (int c = 0; c < NC1; ++c) acc[c] = mfma4(w1r[s5], zb[4 * s5 * LDN + c * 16], acc[c]);
  #pragma unroll
        for (int c = 0; c < NC1; ++c) {
          const int tt = cb1 + c * 16 + l15;
  #pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int co = rt1 * 16 + kq * 4 + reg;
            if (co < C) dxs[co * LDD + tt] = acc[c][reg];        // (+ dy: added row-wise in the copy-out, 8 bytes per read)
          }
        }
      }
      NSC_STAMP(13);
      nsc_lds_barrier();
      NSC_STAMP(14);
      // ---- copy-out: dx rows as whole 256-B lines, 16 bytes per lane (the D-fragment stores were 64-B pieces), . act'(x) ----
      {
        int tid_c = tid;
        asm volatile("" : "+v"(tid_c));
        const int tx = t0 + 4 * (tid_c & 15);
  #pragma unroll
        for (int q = 0; q < NQX; ++q) {
          const int r = (tid_c >> 4) + 32 * q;
          if (r < C && tx < T) {
            f32x4 v = *reinterpret_cast<const f32x4*>(dxs + r * LDD + 4 * (tid_c & 15));
            {   // the residual path: + dy (physical column Hh + 4 + DLT + 4 q4 of the staged row: a multiple of 4, rows 8-byte aligned)
              const float2* yr = reinterpret_cast<const float2*>(dys + r * LDY + Hh + 4 + 4 * (tid_c & 15));
              const float2 y0 = yr[0], y1 = yr[1];
              v[0] += y0.x; v[1] += y0.y; v[2] += y1.x; v[3] += y1.y;
            }
            if (a.in_act == NSC_ACT_LRELU) {
  #pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= (xv[q][e] > 0.f ? 1.f : NSC_LRELU_ALPHA);
            }
            float* gp = a.dx + ((long)b * C + r) * T + tx;
            if constexpr (FIRST) {                           // written through to memory: the pair's second block reads it
              const int bo = ((b * C + r) * T + tx) * 4;
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), sdxo, bo, 0, NSC_AUX_COHERENT);   // (pair launches: T % 4 == 0)
            } else if (tvec) {
              *reinterpret_cast<f32x4*>(gp) = v;
            } else {
  #pragma unroll
              for (int e = 0; e < 4; ++e)
                if (tx + e < T) gp[e] = v[e];
            }
          }
        }
      }
    }
    NSC_STAMP(15);
    nsc_lds_barrier();   // dys / lin / th / dhs / part are rewritten by the next tile
    NSC_STAMP(16);
  }
  NSC_STAMP(17);
}

template <int RT9, int NK9, int DIL, bool CIN1 = false>
__global__ __launch_bounds__(512) void gated_block_dgrad2_kernel(BlockDgradArgs a, int ntiles, int tpf, int skip) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) gated_block_dgrad2_role<RT9, NK9, DIL, CIN1, 0>(a, ntiles, tpf, skip);
  else gated_block_dgrad2_role<RT9, NK9, DIL, CIN1, 1>(a, ntiles, tpf, skip);
}

// the data-path backward of two consecutive blocks of a stack in ONE launch: the dilation-2 block first, then the dilation-1
// block on the dx it wrote (a0.dy must be a1.dx)
// (CIN1B: the dilation-1 block in front has ONE input channel)
template <int RT9, int NK9, bool CIN1B>
__global__ __launch_bounds__(512) void gated_block_dgrad2_pair_kernel(BlockDgradArgs a1, BlockDgradArgs a0, int ntiles, int tpf, int* flags, int* timeouts) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8) == 0) {
    gated_block_dgrad2_role<RT9, NK9, 2, false, 0, false, true>(a1, ntiles, tpf, 0);
    nsc_pair_publish(flags);
    gated_block_dgrad2_role<RT9, NK9, 1, CIN1B, 0, true>(a0, ntiles, tpf, 0, flags, timeouts);
  } else {
    gated_block_dgrad2_role<RT9, NK9, 2, false, 1, false, true>(a1, ntiles, tpf, 0);
    nsc_pair_publish(flags);
    gated_block_dgrad2_role<RT9, NK9, 1, CIN1B, 1, true>(a0, ntiles, tpf, 0, flags, timeouts);
  }
}

template <int RT9, int NK9, int DIL, bool CIN1 = false>
static int launch_block_dgrad2(const BlockDgradArgs& a, hipStream_t st) {
  constexpr int WA16 = ((64 + 14 * DIL + 15) / 16) * 16;
  const size_t partsz = std::max((size_t)4 * NARROW * (WA16 + 4), (size_t)4 * NK9 * 68);
  const int w9t = a.C * 4 + (((a.C * 4) & 15) == 8 ? 0 : 8);
  const size_t smem = ((size_t)4 * NK9 * 110 + (size_t)2 * NARROW * 110 + (size_t)NARROW * 80 + partsz +
                       (size_t)K15 * W15T + (size_t)K9 * w9t) * sizeof(float);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "gated_block_dgrad2: %zu B LDS", smem);
  auto kern = gated_block_dgrad2_kernel<RT9, NK9, DIL, CIN1>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);   // once per instantiation
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_dgrad2: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, 64);
  const int ntiles = a.B * tpf;
  static const int skip = NSC_PROBE_INT("NSC_DGRAD2_SKIP", 0);   // timing probe only
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, 256)), dim3(512), smem, st, a, ntiles, tpf, skip);
  NSC_CHECK_LAUNCH("gated_block_dgrad2");
  return NSC_OK;
}

extern "C" int nsc_gated_block_dgrad(const float* x, const float* h, const float* lin, const float* th, const float* dy,
                                     const float* wt1, const float* wtl, const float* wtr, const float* wt9, float* dx,
                                     float* da, float* dz1, int B, int C, int T, int narrow, int k9, int dil, int in_act,
                                     void* stream) {
  NSC_REQUIRE(x && h && lin && th && dy && wt1 && wtl && wtr && wt9 && dx && da && dz1, NSC_ERR_BAD_ARG,
              "nsc_gated_block_dgrad: null pointer");
  NSC_REQUIRE(B > 0 && C > 1 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad: bad sizes");
  NSC_REQUIRE(narrow == NARROW && k9 == K9 && (dil == 1 || dil == 2) && C <= 112, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_dgrad: built for narrow=20, k9=9, dil in {1,2}, C<=112 (got %d, %d, %d, %d)", narrow, k9, dil, C);
  NSC_REQUIRE(in_act == NSC_ACT_NONE || in_act == NSC_ACT_LRELU, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad: in_act must be none|lrelu");
  const int Hh = 7 * dil, W_a = 64 + 2 * Hh;
  const int nct_a = (W_a + 15) / 16;
  auto ld16 = [](int w) { int l = w; while ((l & 31) != 16) ++l; return l; };
  const int ldy = ld16(nct_a * 16 + 8), lda = ld16(nct_a * 16), ldn = ld16(64);
  const int C4 = (C + 3) & ~3;
  const size_t smem = ((size_t)C4 * ldy + (size_t)3 * NARROW * lda + (size_t)NARROW * ldn) * sizeof(float);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "nsc_gated_block_dgrad: %zu B LDS", smem);
  BlockDgradArgs a{B, C, T, dil, in_act, x, h, lin, th, dy, wt1, wtl, wtr, wt9, dx, da, dz1, da + (long)NARROW * T, 2 * NARROW, nullptr};
  dim3 grid(nsc_cdiv(T, 64), B);
  hipStream_t st = (hipStream_t)stream;
  static const bool v1_only = NSC_PROBE_SET("NSC_BLOCK_DGRAD_V1");   // A/B switch for profiling
  if (!v1_only) {
    if (C == 100) return dil == 1 ? launch_block_dgrad2<7, 25, 1>(a, st) : launch_block_dgrad2<7, 25, 2>(a, st);
    if (C == 50) return dil == 1 ? launch_block_dgrad2<4, 13, 1>(a, st) : launch_block_dgrad2<4, 13, 2>(a, st);
    // C = 25: the K-quarter split of the k9 gradient wants 4 j + 1 channel groups, so the dy tile is staged as 36 rows (9 groups,
    // rows 25..35 zero) on the C = 50 job table
    if (C == 25) return dil == 1 ? launch_block_dgrad2<4, 9, 1>(a, st) : launch_block_dgrad2<4, 9, 2>(a, st);
  }
#define LAUNCH_DG(RT)                                                                                               \
  do {                                                                                                              \
    auto kern = gated_block_dgrad_kernel<RT>;                                                                       \
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);   \
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_dgrad: smem attr: %s", hipGetErrorString(e));         \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, a, ldy, lda, ldn);                                          \
  } while (0)
  if (nsc_cdiv(C, 16) <= 4) LAUNCH_DG(4);
  else LAUNCH_DG(7);
#undef LAUNCH_DG
  NSC_CHECK_LAUNCH("gated_block_dgrad");
  return NSC_OK;
}

// One input channel (first block of a decoder stage): dy [B,C,T] as usual, dx [B,1,T]; wt1 = the [20] taps of the 1 -> 20
// 1x1 conv; dlin / dgate leave as two [B,20,T] tensors (what the per-conv weight-gradient launches of this block read).
extern "C" int nsc_gated_block_dgrad_cin1(const float* h, const float* lin, const float* th, const float* dy,
                                          const float* wt1, const float* wtl, const float* wtr, const float* wt9, float* dx,
                                          float* dlin, float* dgate, float* dz1, int B, int C, int T, int narrow, int k9,
                                          int dil, int da_rows, void* stream) {
  NSC_REQUIRE(h && lin && th && dy && wt1 && wtl && wtr && wt9 && dx && dlin && dgate && dz1, NSC_ERR_BAD_ARG,
              "nsc_gated_block_dgrad_cin1: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_cin1: bad sizes");
  NSC_REQUIRE(da_rows == NARROW || (da_rows == 2 * NARROW && dgate == dlin + (long)NARROW * T), NSC_ERR_BAD_ARG,
              "nsc_gated_block_dgrad_cin1: da_rows must be 20 (two [B,20,T] tensors) or 40 with dgate = dlin + 20 T");
  NSC_REQUIRE(narrow == NARROW && k9 == K9 && (dil == 1 || dil == 2) && (C == 100 || C == 50 || C == 25), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_dgrad_cin1: built for narrow=20, k9=9, dil in {1,2}, C in {100, 50, 25} (got %d, %d, %d, %d)", narrow, k9, dil, C);
  BlockDgradArgs a{B, C, T, dil, NSC_ACT_NONE, dy /* x: unused */, h, lin, th, dy, wt1, wtl, wtr, wt9, dx, dlin, dz1, dgate,
                   da_rows, nullptr};
  hipStream_t st = (hipStream_t)stream;
  if (C == 100) return dil == 1 ? launch_block_dgrad2<7, 25, 1, true>(a, st) : launch_block_dgrad2<7, 25, 2, true>(a, st);
  if (C == 25) return dil == 1 ? launch_block_dgrad2<4, 9, 1, true>(a, st) : launch_block_dgrad2<4, 9, 2, true>(a, st);
  return dil == 1 ? launch_block_dgrad2<4, 13, 1, true>(a, st) : launch_block_dgrad2<4, 13, 2, true>(a, st);
}


// =====================================================================================================
// Kernel-ready parameter IMAGES of a gated block (fast prologue of the persistent kernels).  The caller keeps one image per
// block and direction in device memory and rebuilds it whenever the parameters change with ONE gather (nsc_gather) over
// the index map built here - for the engine that is the launch that flips the data-gradient kernels anyway.
//   which = 0 (forward):  [300][48] LDS image of the k15 gate kernels | per wave and lane: W1 fragments, k9 fragments, biases
//   which = 1 (data gradient): [15][808] | [9][w9t] LDS images | per wave and lane: k9^T fragments, 1x1^T fragments | (Cin = 1: 20 taps)
// Source offsets `offs` (floats, into the buffer the gather reads): which = 0: w1, b1, wl, bl, wr, br, w9, b9;
// which = 1: wt1, wtl, wtr, wt9 (the flipped / transposed kernels of nsc_weight_flip_transpose).  idx[i] = -1: unused pad.
// =====================================================================================================
// rt9 / nk: the template parameters RT9 and NK1 (which = 0) | NK9 (which = 1) of the persistent kernel that serves the shape
static bool img_shape(int which, int C, int Cin, int dil, int* rt9, int* nk) {
  if (!(dil == 1 || dil == 2) || !(C == 100 || C == 50 || C == 25) || !(Cin == C || Cin == 1)) return false;
  *rt9 = C == 100 ? 7 : 4;
  *nk = C == 100 ? 25 : (C == 50 ? 13 : (which == 0 ? 7 : 9));
  return true;
}
extern "C" long nsc_gated_block_image_floats(int which, int C, int Cin, int dil) {
  int rt9, nk;
  if (!img_shape(which, C, Cin, dil, &rt9, &nk)) return 0;
  if (which == 0) {
    // [300][48] k15 gate kernels | (C = 100: [9][20][4] + 4 packed-tile weights, 728) | group A [2][nf4a] | group B [6 | 4][13] |
    // group C [8][2], each x [64 lanes][4]
    const int nk1 = Cin == 1 ? 1 : nk;
    return (long)K15 * NARROW * 48 + (rt9 == 7 ? 728 : 0) + 64L * 4 * (2 * ((nk1 + 4 + 3) / 4) + (rt9 == 7 ? 6 : 4) * 13 + 8 * 2);
  }
  if (which == 1) {
    const int w9t = C * 4 + (((C * 4) & 15) == 8 ? 0 : 8), nj = (nk + 3) / 4 - 1;
    // [15][808] k15 table | [9][w9t] packed k9 table | k9^T fragments, ONE copy per K-quarter [4][nf4s][64][4] | 1x1^T fragments
    // per wave [8][2][64][4] | (Cin = 1: the 20 taps)
    return (long)K15 * W15T + (long)K9 * w9t + 64L * 4 * (4 * ((K9 * nj + 3 + 3) / 4) + 8 * 2) + (Cin == 1 ? NARROW : 0);
  }
  return 0;
}
extern "C" int nsc_gated_block_image_index(int which, int C, int Cin, int dil, const long* offs, int* idx) {
  NSC_REQUIRE(offs && idx, NSC_ERR_BAD_ARG, "nsc_gated_block_image_index: null pointer");
  int rt9, nk;
  NSC_REQUIRE((which == 0 || which == 1) && img_shape(which, C, Cin, dil, &rt9, &nk), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_image_index: no image for C %d, Cin %d, dil %d, which %d", C, Cin, dil, which);
  const long n = nsc_gated_block_image_floats(which, C, Cin, dil);
  for (long i = 0; i < n; ++i) idx[i] = -1;
  auto mn = [](int a, int b) { return a < b ? a : b; };
  if (which == 0) {
    const long w1 = offs[0], b1 = offs[1], wl = offs[2], bl = offs[3], wr = offs[4], br = offs[5], w9 = offs[6], b9 = offs[7];
    const int nk1 = Cin == 1 ? 1 : nk, LDW = 48;
    for (int e = 0; e < K15 * NARROW * LDW; ++e) {
      const int row = e / LDW, r = e - row * LDW, ii = r & 15;
      const int c = mn((r >> 4) * 8 + (ii >> 2) * 2 + (ii & 1), NARROW - 1);
      idx[e] = (int)(((ii & 2) ? wr : wl) + row * NARROW + c);
    }
    const long baseP = (long)K15 * NARROW * LDW;
    if (rt9 == 7) {          // packed tile of channels 96..99: [9][20][4] weights, 4 biases
      for (int e = 0; e < K9 * NARROW * 4; ++e) idx[baseP + e] = (int)(w9 + (long)(e >> 2) * C + 96 + (e & 3));
      for (int e = 0; e < 4; ++e) idx[baseP + K9 * NARROW * 4 + e] = (int)(b9 + 96 + e);
    }
    const int nfa = nk1 + 4, nf4a = (nfa + 3) / 4, nf4b = 13, nf4c = 2, nvb = rt9 == 7 ? 6 : 4;
    const long baseA = baseP + (rt9 == 7 ? 728 : 0), baseB = baseA + 2L * nf4a * 256, baseC = baseB + (long)nvb * nf4b * 256;
    for (int lane = 0; lane < 64; ++lane) {
      const int l15 = lane & 15, kq = lane >> 4;
      for (int r1 = 0; r1 < 2; ++r1)             // group A: a wave's row tile of h (waves 0-3: channels 0..15, waves 4-7: 16..19 + padding)
        for (int f = 0; f < nfa; ++f) {
          const long src = f < nk1 ? w1 + (long)mn(4 * f + kq, Cin - 1) * NARROW + mn(r1 * 16 + l15, NARROW - 1)
                                   : b1 + mn(r1 * 16 + kq * 4 + (f - nk1), NARROW - 1);
          idx[baseA + ((long)(r1 * nf4a + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
        }
      for (int rt3 = 0; rt3 < nvb; ++rt3)        // group B: a phase-3 row tile (gated_block_fwd2_kernel's job table)
        for (int f = 0; f < 49; ++f) {
          long src;
          if (f < 45) { const int tap = f / 5, u = f % 5; src = w9 + (long)(tap * NARROW + 4 * u + kq) * C + mn(rt3 * 16 + l15, C - 1); }
          else src = b9 + mn(rt3 * 16 + kq * 4 + (f - 45), C - 1);
          idx[baseB + ((long)(rt3 * nf4b + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
        }
      for (int wave = 0; wave < 8; ++wave) {     // group C: gate biases of the wave's two phase-2 jobs
        int jrt[2];
        for (int e = 0; e < 2; ++e) { const int q = wave + 8 * e; jrt[e] = (q < 15 ? q : wave) % 3; }
        for (int f = 0; f < 8; ++f) {
          const int t = f & 3;
          const long src = (f < 4 ? bl : br) + mn(jrt[t >> 1] * 8 + kq * 2 + (t & 1), NARROW - 1);
          idx[baseC + ((long)(wave * nf4c + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
        }
      }
    }
    return NSC_OK;
  }
  const long wt1 = offs[0], wtl = offs[1], wtr = offs[2], wt9 = offs[3];
  const int w9t = C * 4 + (((C * 4) & 15) == 8 ? 0 : 8), nj = (nk + 3) / 4 - 1;
  for (int e = 0; e < K15 * 2 * NARROW * NARROW; ++e) {
    const int tp = e / (2 * NARROW * NARROW), r = e - tp * 2 * NARROW * NARROW, cp = r / NARROW, ci = r - cp * NARROW;
    idx[e + 8 * tp] = (int)((cp < NARROW ? wtl : wtr) + (long)(tp * NARROW + (cp < NARROW ? cp : cp - NARROW)) * NARROW + ci);
  }
  const long base9 = (long)K15 * W15T;
  for (int e = 0; e < K9 * C * 4; ++e)
    idx[base9 + (long)(e / (4 * C)) * w9t + (e % (4 * C))] = (int)(wt9 + (long)(e >> 2) * NARROW + 16 + (e & 3));
  const long baseB = base9 + (long)K9 * w9t;
  const int nfs = K9 * nj + 3, nf4s = (nfs + 3) / 4, nf4w = 2;
  // k9^T fragments: waves kg and kg + 4 share a K-quarter and read the same copy
  for (int kg = 0; kg < 4; ++kg)
    for (int lane = 0; lane < 64; ++lane) {
      const int l15 = lane & 15, kq = lane >> 4;
      for (int f = 0; f < nfs; ++f) {
        long src;
        if (f < K9 * nj) { const int tp = f / nj, j = f % nj; src = wt9 + ((long)tp * C + mn(4 * (kg + 4 * j) + kq, C - 1)) * NARROW + l15; }
        else { const int i = f - K9 * nj; src = wt9 + ((long)mn(kg + 4 * i, K9 - 1) * C + mn(4 * (nk - 1) + kq, C - 1)) * NARROW + l15; }
        idx[baseB + ((long)(kg * nf4s + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
      }
    }
  const long baseW = baseB + 4L * nf4s * 64 * 4;
  if (Cin != 1)                                            // (the 1x1 gradient of a one-channel input is a dot product)
    for (int wave = 0; wave < 8; ++wave) {
      const int rt1 = rt9 == 7 ? mn(wave, 6) : (wave & 3);
      for (int lane = 0; lane < 64; ++lane) {
        const int l15 = lane & 15, kq = lane >> 4;
        for (int s5 = 0; s5 < 5; ++s5)
          idx[baseW + ((long)(wave * nf4w + s5 / 4) * 64 + lane) * 4 + (s5 & 3)] = (int)(wt1 + (long)(s5 * 4 + kq) * C + mn(rt1 * 16 + l15, C - 1));
      }
    }
  if (Cin == 1)
    for (int c = 0; c < NARROW; ++c) idx[baseW + 8L * nf4w * 64 * 4 + c] = (int)(wt1 + c);
  return NSC_OK;
}

// ---- pair launches: two consecutive blocks of a stack (dilations 1, 2; C -> C both) in one launch ----
static int nsc_cu_count() {
  static std::atomic<int> cached[64];
  int dv = 0;
  if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= 64) return 0;
  int n = cached[dv].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dv) != hipSuccess) return 0;
    cached[dv].store(n, std::memory_order_relaxed);
  }
  return n;
}
extern "C" int nsc_gated_block_pair_flag_ints(void) { return NSC_PAIR_MAX_WG; }     // one flag per workgroup of a pair launch

template <int RT9, int NK1A, int NK1B>
static int launch_block_fwd2_pair(const BlockArgs& a0, const BlockArgs& a1, int* flags, int* timeouts, hipStream_t st) {
  constexpr int CR = 4 * (NK1A > NK1B ? NK1A : NK1B);
  const size_t smem = ((size_t)(CR + NARROW) * 112 + (size_t)3 * NARROW * 80 + (size_t)K15 * NARROW * 48 + (RT9 == 7 ? 728 : 0)) * sizeof(float);
  auto kern = gated_block_fwd2_pair_kernel<RT9, NK1A, NK1B>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd2_pair: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a0.T, 64);
  const int ntiles = a0.B * tpf;
  const int grid = std::min(ntiles, 256);
  // every workgroup of the launch must be resident at once (they wait for each other): one per CU (LDS), so grid <= CUs
  NSC_REQUIRE(grid <= nsc_cu_count(), NSC_ERR_UNSUPPORTED, "gated_block_fwd2_pair: %d workgroups > %d CUs", grid, nsc_cu_count());
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a0, a1, ntiles, tpf, flags, timeouts);
  NSC_CHECK_LAUNCH("gated_block_fwd2_pair");
  return NSC_OK;
}

extern "C" int nsc_gated_block_pair_fwd_img(const float* img0, const float* img1, const float* x, float* out0, float* h0, float* lin0,
                                            float* th0, float* g0, float* out1, float* h1, float* lin1, float* th1, float* g1, int B,
                                            int C, int Cin0, int T, int flat1, int* flags, int* timeouts, void* stream) {
  NSC_REQUIRE(img0 && img1 && x && out0 && out1 && flags && timeouts, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_img: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_img: bad sizes");
  NSC_REQUIRE(C == 100 || C == 50 || C == 25, NSC_ERR_UNSUPPORTED, "nsc_gated_block_pair_fwd_img: C %d", C);
  NSC_REQUIRE((T & 3) == 0 && (long)B * C * T * 4 < (1L << 31), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_pair_fwd_img: needs T %% 4 == 0 and a tensor below 2 GB (T %d, B %d): launch the blocks one by one", T, B);
  NSC_REQUIRE((((uintptr_t)img0 | (uintptr_t)img1) & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_img: images must be 16-byte aligned");
  NSC_REQUIRE((!(lin0 || th0 || g0) || (lin0 && th0 && g0)) && (!(lin1 || th1 || g1) || (lin1 && th1 && g1)), NSC_ERR_BAD_ARG,
              "nsc_gated_block_pair_fwd_img: lin/th/g outputs must be given together");
  NSC_REQUIRE(Cin0 == C || Cin0 == 1, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_img: Cin0 must be C or 1 (got %d)", Cin0);
  BlockArgs a0{B, C, T, 1, 0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out0, h0, lin0, th0, g0, Cin0, img0};
  BlockArgs a1{B, C, T, 2, flat1, out0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out1, h1, lin1, th1, g1, C, img1};
  hipStream_t st = (hipStream_t)stream;
  if (Cin0 == 1) {
    if (C == 100) return launch_block_fwd2_pair<7, 1, 25>(a0, a1, flags, timeouts, st);
    if (C == 25) return launch_block_fwd2_pair<4, 1, 7>(a0, a1, flags, timeouts, st);
    return launch_block_fwd2_pair<4, 1, 13>(a0, a1, flags, timeouts, st);
  }
  if (C == 100) return launch_block_fwd2_pair<7, 25, 25>(a0, a1, flags, timeouts, st);
  if (C == 25) return launch_block_fwd2_pair<4, 7, 7>(a0, a1, flags, timeouts, st);
  return launch_block_fwd2_pair<4, 13, 13>(a0, a1, flags, timeouts, st);
}

template <int RT9, int NK9, bool CIN1B>
static int launch_block_dgrad2_pair(const BlockDgradArgs& a1, const BlockDgradArgs& a0, int* flags, int* timeouts, hipStream_t st) {
  size_t smem = 0;
  for (int dil = 1; dil <= 2; ++dil) {             // the larger of the two bodies' LDS layouts (dilation 2)
    const int WA16 = ((64 + 14 * dil + 15) / 16) * 16;
    const size_t partsz = std::max((size_t)4 * NARROW * (WA16 + 4), (size_t)4 * NK9 * 68);
    const int w9t = a0.C * 4 + (((a0.C * 4) & 15) == 8 ? 0 : 8);
    smem = std::max(smem, ((size_t)4 * NK9 * 110 + (size_t)2 * NARROW * 110 + (size_t)NARROW * 80 + partsz + (size_t)K15 * W15T +
                           (size_t)K9 * w9t) * sizeof(float));
  }
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "gated_block_dgrad2_pair: %zu B LDS", smem);
  auto kern = gated_block_dgrad2_pair_kernel<RT9, NK9, CIN1B>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_dgrad2_pair: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a0.T, 64);
  const int ntiles = a0.B * tpf;
  const int grid = std::min(ntiles, 256);
  NSC_REQUIRE(grid <= nsc_cu_count(), NSC_ERR_UNSUPPORTED, "gated_block_dgrad2_pair: %d workgroups > %d CUs", grid, nsc_cu_count());
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a1, a0, ntiles, tpf, flags, timeouts);
  NSC_CHECK_LAUNCH("gated_block_dgrad2_pair");
  return NSC_OK;
}

// block 1 = the dilation-2 block (runs first: dy1 in, dx1 out), block 0 = the dilation-1 block in front of it (its dy is dx1)
extern "C" int nsc_gated_block_pair_dgrad_img(const float* img1, const float* x1, const float* h1, const float* lin1, const float* th1,
                                              const float* dy1, float* dx1, float* da1, float* dz1_1, const float* img0,
                                              const float* x0, const float* h0, const float* lin0, const float* th0, float* dx0,
                                              float* da0, float* dz1_0, int B, int C, int Cin0, int T, int in_act0, int* flags,
                                              int* timeouts, void* stream) {
  NSC_REQUIRE(img1 && x1 && h1 && lin1 && th1 && dy1 && dx1 && da1 && dz1_1 && img0 && (x0 || Cin0 == 1) && h0 && lin0 && th0 && dx0 && da0 &&
                  dz1_0 && flags && timeouts, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_dgrad_img: null pointer");
  NSC_REQUIRE(Cin0 == C || (Cin0 == 1 && in_act0 == NSC_ACT_NONE), NSC_ERR_BAD_ARG,
              "nsc_gated_block_pair_dgrad_img: Cin0 must be C, or 1 with in_act0 none (got %d, %d)", Cin0, in_act0);
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_dgrad_img: bad sizes");
  NSC_REQUIRE(C == 100 || C == 50 || C == 25, NSC_ERR_UNSUPPORTED, "nsc_gated_block_pair_dgrad_img: C %d", C);
  NSC_REQUIRE((T & 3) == 0 && (long)B * C * T * 4 < (1L << 31), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_pair_dgrad_img: needs T %% 4 == 0 and a tensor below 2 GB (T %d, B %d): launch the blocks one by one", T, B);
  NSC_REQUIRE((((uintptr_t)img0 | (uintptr_t)img1) & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_dgrad_img: images must be 16-byte aligned");
  NSC_REQUIRE(in_act0 == NSC_ACT_NONE || in_act0 == NSC_ACT_LRELU, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_dgrad_img: in_act0");
  // the dilation-2 block's input is the dilation-1 block's lrelu output: its dx carries lrelu'(x1)
  BlockDgradArgs a1{B, C, T, 2, NSC_ACT_LRELU, x1, h1, lin1, th1, dy1, nullptr, nullptr, nullptr, nullptr, dx1, da1, dz1_1,
                    da1 + (long)NARROW * T, 2 * NARROW, img1};
  BlockDgradArgs a0{B, C, T, 1, in_act0, Cin0 == 1 ? dx1 : x0, h0, lin0, th0, dx1, nullptr, nullptr, nullptr, nullptr, dx0, da0, dz1_0,
                    da0 + (long)NARROW * T, 2 * NARROW, img0};
  hipStream_t st = (hipStream_t)stream;
  if (Cin0 == 1) {
    if (C == 100) return launch_block_dgrad2_pair<7, 25, true>(a1, a0, flags, timeouts, st);
    if (C == 25) return launch_block_dgrad2_pair<4, 9, true>(a1, a0, flags, timeouts, st);
    return launch_block_dgrad2_pair<4, 13, true>(a1, a0, flags, timeouts, st);
  }
  if (C == 100) return launch_block_dgrad2_pair<7, 25, false>(a1, a0, flags, timeouts, st);
  if (C == 25) return launch_block_dgrad2_pair<4, 9, false>(a1, a0, flags, timeouts, st);
  return launch_block_dgrad2_pair<4, 13, false>(a1, a0, flags, timeouts, st);
}

// The two persistent kernels on an image (shapes of the codec only: C in {100, 50, 25}, Cin in {C, 1}, dil in {1, 2}).
extern "C" int nsc_gated_block_fwd_img(const float* img, const float* x, float* out, float* h_out, float* lin_out, float* th_out,
                                       float* g_out, int B, int C, int Cin, int T, int dil, int flat, void* stream) {
  NSC_REQUIRE(img && x && out, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_img: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_img: bad sizes");
  NSC_REQUIRE(nsc_gated_block_image_floats(0, C, Cin, dil) > 0, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_fwd_img: no image kernel for C %d, Cin %d, dil %d", C, Cin, dil);
  NSC_REQUIRE(((uintptr_t)img & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_img: image must be 16-byte aligned");
  NSC_REQUIRE(!(lin_out || th_out || g_out) || (lin_out && th_out && g_out), NSC_ERR_BAD_ARG,
              "nsc_gated_block_fwd_img: lin/th/g outputs must be given together");
  BlockArgs a{B, C, T, dil, flat, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, h_out, lin_out,
              th_out, g_out, Cin, img};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 1) {
    if (C == 100) return dil == 1 ? launch_block_fwd2<7, 1, 1>(a, st) : launch_block_fwd2<7, 1, 2>(a, st);
    return dil == 1 ? launch_block_fwd2<4, 1, 1>(a, st) : launch_block_fwd2<4, 1, 2>(a, st);
  }
  if (C == 100) return dil == 1 ? launch_block_fwd2<7, 25, 1>(a, st) : launch_block_fwd2<7, 25, 2>(a, st);
  if (C == 25) return dil == 1 ? launch_block_fwd2<4, 7, 1>(a, st) : launch_block_fwd2<4, 7, 2>(a, st);
  return dil == 1 ? launch_block_fwd2<4, 13, 1>(a, st) : launch_block_fwd2<4, 13, 2>(a, st);
}

extern "C" int nsc_gated_block_dgrad_img(const float* img, const float* x, const float* h, const float* lin, const float* th,
                                         const float* dy, float* dx, float* dlin, float* dgate, float* dz1, int B, int C, int Cin,
                                         int T, int dil, int in_act, int da_rows, void* stream) {
  NSC_REQUIRE(img && h && lin && th && dy && dx && dlin && dgate && dz1 && (x || Cin == 1), NSC_ERR_BAD_ARG,
              "nsc_gated_block_dgrad_img: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_img: bad sizes");
  NSC_REQUIRE(nsc_gated_block_image_floats(1, C, Cin, dil) > 0, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_dgrad_img: no image kernel for C %d, Cin %d, dil %d", C, Cin, dil);
  NSC_REQUIRE(((uintptr_t)img & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_img: image must be 16-byte aligned");
  NSC_REQUIRE(in_act == NSC_ACT_NONE || (in_act == NSC_ACT_LRELU && Cin > 1), NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_img: in_act");
  NSC_REQUIRE(da_rows == NARROW || (da_rows == 2 * NARROW && dgate == dlin + (long)NARROW * T), NSC_ERR_BAD_ARG,
              "nsc_gated_block_dgrad_img: da_rows must be 20 (two [B,20,T] tensors) or 40 with dgate = dlin + 20 T");
  BlockDgradArgs a{B, C, T, dil, in_act, Cin == 1 ? dy : x, h, lin, th, dy, nullptr, nullptr, nullptr, nullptr, dx, dlin, dz1, dgate,
                   da_rows, img};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 1) {
    if (C == 100) return dil == 1 ? launch_block_dgrad2<7, 25, 1, true>(a, st) : launch_block_dgrad2<7, 25, 2, true>(a, st);
    if (C == 25) return dil == 1 ? launch_block_dgrad2<4, 9, 1, true>(a, st) : launch_block_dgrad2<4, 9, 2, true>(a, st);
    return dil == 1 ? launch_block_dgrad2<4, 13, 1, true>(a, st) : launch_block_dgrad2<4, 13, 2, true>(a, st);
  }
  if (C == 100) return dil == 1 ? launch_block_dgrad2<7, 25, 1>(a, st) : launch_block_dgrad2<7, 25, 2>(a, st);
  if (C == 25) return dil == 1 ? launch_block_dgrad2<4, 9, 1>(a, st) : launch_block_dgrad2<4, 9, 2>(a, st);
  return dil == 1 ? launch_block_dgrad2<4, 13, 1>(a, st) : launch_block_dgrad2<4, 13, 2>(a, st);
}
