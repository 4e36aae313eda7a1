// Data-path backward of the gated block (nn_core_operator.py:82-112) on the bf16 matrix cores with SPLIT OPERANDS, as two or three
// launches per block instead of one fused persistent kernel:
//   bb_gemm_kernel<50, 9, 1, 2, 2 | 0>  dg = W9^T * dy over 50 channels of dy at a time (C = 100: a partial sum, then the rest), the GLU
//                                       backward in the last one's epilogue                           -> da [B,40,T] = dlin | dgate
//   bb_gemm_kernel<40, 15, D, 2, 3>     dh = Wl^T dlin + Wr^T dgate, . lrelu'(h) -> dz1 [B,20,T]; then, from the dz1 tile in LDS,
//                                       dx = (W1^T dz1 + dy) . act'(x) on the vector ALU (HBM-bound)   -> dx [B,C,T]
//   (one input channel: EPI 1 and bb_1x1_cin1_kernel)
// Why not fused (block.hip: gated_block_dgrad2, block_split.hip: gated_block_dgrad3): the two long contractions have 20 output rows.
// On 16-row matrix tiles they pad to 32, the fused kernels recompute a halo of 7 * dil columns of dg per 64-step tile, and their
// weights (the same for every tile) only fit the registers when the REDUCTION is split over the eight waves - an eight-way partial
// sum per output.  Here every contraction is written as a polyphase GEMM with 80 = 4 x 20 rows (exactly five row tiles):
//   out[ci, 4 n + p] = sum_{tap, o} W[tap][ci][o] . in[o, 4 n + p + half - tap]        (conv^T of a K-tap SAME conv, half = (K-1)/2)
//                    = sum_{j, o} Wm[(p, ci)][(j, o)] . X[(j, o)][n]      with  X[(j, o)][n] = in[o, 4 n - half + j],  j = 0 .. K + 2,
//                                                                               Wm[(p, ci)][(j, o)] = W[p + K - 1 - j][ci][o]  (0 outside)
// i.e. a stride-4 conv with K + 3 window rows over a [time][channels] plane: 4/3 (9 taps) resp. 6/5 (15 taps) of the products, no
// padded rows, no halo recompute, one accumulator tile per 16 x 4 outputs.  Dilation 2 = the same GEMM on the two parity subsequences.
// Layout of a tile in LDS: the activation plane is kept as PHASE planes [r mod 4][r / 4][CP] (r = row of the subsequence, CP = channels
// rounded to 8), so that the 16 columns of a column tile (rows 4 n + j) are 16 consecutive rows of ONE phase plane: a fragment of 8
// consecutive k = (j, o .. o + 7) is one aligned ds_read_b128 and the 16 lanes of a quarter wave are CP * 2 bytes apart (208 | 112 | 80:
// conflict-free).  The WEIGHTS are not inflated: the image is the three bf16 pieces of the kernel in its own layout [tap][ci][CP]
// (nsc_gated_block_simage_index, which = 2) and a lane computes the address of its fragment (tap = p + K - 1 - j; a tap outside the
// kernel reads 16 bytes of zeros kept behind each plane): 56 KB (C = 100: W9) + 36 KB (Wl | Wr) per block, L2-resident.
// A workgroup (8 waves) owns a tile of 16 NC columns x 4 phases (x dilation) = 64 NC time steps: stage (fp32 [ch][time] -> three bf16
// pieces), GEMM, combine in an fp32 tile [20][steps] in LDS, elementwise epilogue with 16-byte coalesced stores.  The 5 row tiles x KS
// k-steps are cut into four equal contiguous runs, one per SIMD (its two waves take the two rows the run touches): every SIMD issues the
// same number of MFMAs and a row tile has at most two partial sums (first: store, second: add - deterministic).
#include "nsc_common.h"
#include <algorithm>

#include "block_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#ifdef NSC_PROBES
extern "C" int nsc_probe_read_bb(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsc_dbg_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3;
}
#endif

namespace {
__device__ __forceinline__ f32x4 bb_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int CH_, int K_, int DIL_, int NC_>
struct BbGeom {
  static constexpr int CH = CH_, CP = (CH_ + 7) & ~7, G8 = CP / 8;      // channels of the activation, plane pitch, 16-byte groups per row
  static constexpr int K = K_, HALF = (K_ - 1) / 2, KT = K_ + 3;        // window rows per column that carry weights
  static constexpr int KS = (KT * CP + 31) / 32;                        // k-steps of 32
  static constexpr int JMAX = (KS * 32 + CP - 1) / CP;                  // window rows a column touches (the rows past KT meet zero weights)
  static constexpr int D = DIL_, NC = NC_, NCOL = 16 * NC_, NCS = NCOL / DIL_;   // columns per tile / per subsequence
  static constexpr int TS = 4 * NCOL;                                   // time steps per tile
  static constexpr int RI = NCS + (JMAX + 3) / 4;                       // rows of a phase plane (i = n + j / 4)
  static constexpr int PLS = DIL_ * 4 * RI * CP;                        // elements per piece: [subsequence][phase][RI][CP]
  static constexpr int NREL = 4 * RI * DIL_;                            // time steps staged per tile, from t0 - D * HALF
  static constexpr int SH = (4 - (DIL_ * HALF) % 4) % 4;                // the first 16-byte group starts SH steps before that
  static constexpr int NG = (SH + NREL + 3) / 4;                        // 16-byte groups per channel row
  static constexpr int OP = TS + 4;                                     // row pitch (floats) of the combined tile [20][OP]
  static constexpr int PLW = K_ * NARROW * CP + 8;                      // elements per piece of the weight image (8 zeros behind)
  static constexpr int ZOFF = K_ * NARROW * CP;
  static constexpr int NSLOT = 4;                                       // partial-sum tiles [20][OP]: waves w and w + 4 share slot w
  static constexpr size_t smem = (size_t)3 * PLW * 2 + (size_t)3 * PLS * 2 + (size_t)NSLOT * NARROW * OP * 4;
  static_assert(PLS % 8 == 0 && PLW % 8 == 0 && NC_ % DIL_ == 0 && CP % 8 == 0, "16-byte aligned planes");
};

struct BbArgs {
  const float* src;      // the activation: dy [B,src_C,T] (channels c0 .. c0 + CH - 1 of it) | da [B,40,T]
  const u16* img;        // weight pieces [3][PLW]
  const float* e0;       // epilogue operands [B,20,T]: lin | h
  const float* e1;       //                             th  | -
  const float* pin;      // nullable: a partial sum [B,20,T] added before the epilogue (the other half of the channels)
  float* out;            // da [B,40,T] | dz1 [B,20,T] | the partial sum [B,20,T]
  int B, T, src_C, c0, ntiles, tpf;
  // EPI 3 (the k15 launch with the 1x1 gradient behind it): dx [B,C1,T] = (W1^T dz1 + dy) . act'(x)
  const float *w1, *dy, *x;
  float* dx;
  int C1, in_act;
};

// EPI 0: out = (v . th | v . lin . (1 - th^2));  EPI 1: out = v . lrelu'(h);  EPI 2: out = v (a partial sum over half of the channels);
// EPI 3: EPI 1, then dx = (W1^T dz1 + dy) . act'(x) from the dz1 tile in LDS (the HBM-bound 1x1 launch of its own was 18 us per block)
template <int CH_, int K_, int DIL_, int NC_, int EPI>
__global__ __launch_bounds__(512) void bb_gemm_kernel(BbArgs a) {
  using G = BbGeom<CH_, K_, DIL_, NC_>;
  constexpr int CH = G::CH, CP = G::CP, G8 = G::G8, K = G::K, KS = G::KS, D = G::D, NC = G::NC, TS = G::TS, RI = G::RI, PLS = G::PLS;
  constexpr int OP = G::OP, PLW = G::PLW;
  extern __shared__ __attribute__((aligned(16))) u16 bb_sm[];
  u16* const wts = bb_sm;                                               // [3][PLW]   the weight pieces, resident for the whole launch
  u16* const planes = bb_sm + 3 * PLW;                                  // [3][PLS]
  float* const outp = reinterpret_cast<float*>(planes + 3 * PLS);       // [4][20][OP]  partial-sum slots
  float* const w1s = outp + G::NSLOT * NARROW * OP;                     // EPI 3: [C1][20] the 1x1 kernel (parameter layout)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int T = a.T;

  // ---- this wave's share of the GEMM: ALL five row tiles x all column tiles over its eighth of the k-steps.  Every weight and every
  // activation fragment is then read from LDS exactly once per tile (a wave that owned one row tile over many k-steps re-read the
  // activations five times: 95 B/clk of LDS reads, twice the MFMA time); the price is eight partial sums per output, met in LDS. ----
  const int ks0 = wave * KS / 8, ks1 = (wave + 1) * KS / 8;
  // A side (weights): row m = 16 r + l15 = 20 p + ci; the piece row of (m, window row j) is (K - 1 - j) * 20 + m - linear in m: the 16
  // lanes of a quarter wave read 16 consecutive rows (CP * 2 bytes apart) - where the tap p + K - 1 - j exists, else the zeros
  int tapb[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) tapb[r] = (16 * r + l15) / NARROW + K - 1;
  const int abase = ((K - 1) * NARROW + l15) * CP;
  const u16* const bcol = planes + l15 * CP;

  // ---- staging: a unit = (channel pair, 16-byte group of 4 steps); the loads of a tile are issued a whole tile ahead ----
  constexpr int NPAIR = (CH + 1) / 2, NCPB = (NPAIR + 7) / 8, NGB = (G::NG + 7) / 8, NIT = (NCPB * NGB + 7) / 8;
  const int gl = lane & 7, cl = lane >> 3;
  f32x4 v0[NIT], v1[NIT];
  auto load_tile = [&](int tile) {
    const int b = tile / a.tpf, t0 = (tile - b * a.tpf) * TS;
    const float* xb = a.src + ((long)b * a.src_C + a.c0) * T;
    const int t_al = t0 - D * G::HALF - G::SH;                            // multiple of 4
#pragma unroll
    for (int it = 0; it < NIT; ++it) {                                    // branch-free: clamped address, select on the value
      const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
      const int cp = cpb * 8 + cl, g = gb * 8 + gl, t = t_al + 4 * g;
      const bool ok = ub < NCPB * NGB && cp < NPAIR && g < G::NG && t >= 0 && t < T;
      const bool ok1 = ok && 2 * cp + 1 < CH;
      const f32x4 l0 = *reinterpret_cast<const f32x4*>(xb + (ok ? (long)(2 * cp) * T + t : 0));
      const f32x4 l1 = *reinterpret_cast<const f32x4*>(xb + (ok1 ? (long)(2 * cp + 1) * T + t : 0));
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v0[it] = ok ? l0 : z;
      v1[it] = ok1 ? l1 : z;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
      const int cp = cpb * 8 + cl, g = gb * 8 + gl;
      if (ub < NCPB * NGB && cp < NPAIR && g < G::NG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rel = 4 * g + i - G::SH;
          if (rel >= 0 && rel < G::NREL) {
            const int e = D == 1 ? 0 : (rel & 1), r = D == 1 ? rel : (rel >> 1);
            unsigned pk[3];
            nsc_split2(v0[it][i], v1[it][i], pk);
            unsigned* w = reinterpret_cast<unsigned*>(planes + ((e * 4 + (r & 3)) * RI + (r >> 2)) * CP + 2 * cp);
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p * (PLS / 2)] = pk[p];
          }
        }
      }
    }
  };
  // ---- once per workgroup: the first tile's loads go out first; then the weight pieces (L2 -> LDS: all of a thread's 16-byte loads in
  // flight at once - a load/store loop was eight serial round trips, 9 us per launch) and zeros in the planes (their pad channels
  // [CH, CP) are read against zero weights and never written by the staging) ----
  if ((int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);
  if (EPI == 3)
    for (int e = tid; e < a.C1 * NARROW / 4; e += 512) reinterpret_cast<f32x4*>(w1s)[e] = reinterpret_cast<const f32x4*>(a.w1)[e];
  {
    constexpr int NW16 = 3 * PLW / 8, NWI = (NW16 + 511) / 512;
    uint4 wv[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) wv[i] = reinterpret_cast<const uint4*>(a.img)[min(tid + 512 * i, NW16 - 1)];
    for (int e = tid; e < 3 * PLS / 8; e += 512) reinterpret_cast<uint4*>(planes)[e] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < NWI; ++i)
      if (tid + 512 * i < NW16) reinterpret_cast<uint4*>(wts)[tid + 512 * i] = wv[i];
  }
  __syncthreads();

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tpf, t0 = (tile - b * a.tpf) * TS;
    NSC_STAMP(0);
    // ---- the epilogue's operands: requested first, consumed last ----
    constexpr int NEI = (NARROW * (TS / 4) + 511) / 512;
    f32x4 ev0[NEI], ev1[NEI], pv[NEI];
#pragma unroll
    for (int it = 0; it < NEI; ++it) pv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NEI; ++it) {
      const int id = tid + 512 * it, c = id / (TS / 4), tg = id - c * (TS / 4), t = t0 + 4 * tg;
      const bool ok = c < NARROW && t < T;
      const long off = ((long)b * NARROW + (ok ? c : 0)) * T + (ok ? t : 0);
      if (EPI != 2) ev0[it] = *reinterpret_cast<const f32x4*>(a.e0 + off);       // (EPI 1 | 3: h)
      if (EPI == 0) ev1[it] = *reinterpret_cast<const f32x4*>(a.e1 + off);
      if (EPI == 0 && a.pin) pv[it] = *reinterpret_cast<const f32x4*>(a.pin + off);
    }
    store_tile();
    NSC_STAMP(1);
    __syncthreads();
    NSC_STAMP(2);
    if (tile + (int)gridDim.x < a.ntiles) load_tile(tile + gridDim.x);     // in flight during the GEMM (which reads LDS only)

    // ---- GEMM: units (k-step s, row tile r); the next unit's weight fragment and, at r = 4, the next step's activation fragments are
    // requested before the unit's 6 NC products ----
    f32x4 acc[5][NC];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int e = 0; e < NC; ++e) acc[r][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      bf16x8 af[2][3], bf[2][NC][3];
      auto load_a = [&](int s, int r, bf16x8 (&dst)[3]) {
        const int kg = 4 * s + q, j = kg / G8, o8 = kg - j * G8;
        const int off = (unsigned)(tapb[r] - j) < (unsigned)K ? abase + (16 * r - j * NARROW) * CP + 8 * o8 : G::ZOFF;
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = *reinterpret_cast<const bf16x8*>(wts + p * PLW + off);
      };
      auto load_b = [&](int s, bf16x8 (&dst)[NC][3]) {
        const int kg = 4 * s + q, j = kg / G8, o8 = kg - j * G8;
        const u16* pr = bcol + ((j & 3) * RI + (j >> 2)) * CP + 8 * o8;
#pragma unroll
        for (int e = 0; e < NC; ++e) {
          constexpr int NCH = NC / D;                                    // column tiles per subsequence
          const int cb = ((e / NCH) * 4 * RI + (e % NCH) * 16) * CP;
#pragma unroll
          for (int p = 0; p < 3; ++p) dst[e][p] = *reinterpret_cast<const bf16x8*>(pr + p * PLS + cb);
        }
      };
      const int last = ks1 - 1;
      load_b(ks0, bf[0]);
      load_a(ks0, 0, af[0]);
#pragma unroll 1
      for (int s = ks0; s < ks1; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int sc = s + u, sn = min(sc + 1, last);
#pragma unroll
          for (int r = 0; r < 5; ++r) {
            constexpr int dummy = 0; (void)dummy;
            const int cur = (5 * u + r) & 1;
            if (r < 4) {
              load_a(min(sc, last), r + 1, af[cur ^ 1]);
            } else {
              load_a(sn, 0, af[cur ^ 1]);
              load_b(sn, bf[(u + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sc < ks1) {
              constexpr int PW[6] = {1, 0, 2, 0, 1, 0}, PX[6] = {1, 2, 0, 1, 0, 0};    // (weight piece, activation piece), smallest first
#pragma unroll
              for (int pi = 0; pi < 6; ++pi)
#pragma unroll
                for (int e = 0; e < NC; ++e) acc[r][e] = bb_mfma(af[cur][PW[pi]], bf[u & 1][e][PX[pi]], acc[r][e]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    NSC_STAMP(3);
    // ---- combine: waves 0..3 store their partial sums into slots 0..3, waves 4..7 add theirs; the epilogue adds the four slots ----
    {
      float* const slot = outp + (wave & 3) * (NARROW * OP);
      auto flush = [&](bool add) {
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
          for (int e = 0; e < NC; ++e) {
            constexpr int NCH = NC / D;
            const int sub = e / NCH, n = (e % NCH) * 16 + l15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int mm = 16 * r + 4 * q + i, pm_ = mm / NARROW, ci_ = mm - pm_ * NARROW;
              float* dst = slot + ci_ * OP + sub + D * (4 * n + pm_);
              *dst = add ? *dst + acc[r][e][i] : acc[r][e][i];
            }
          }
      };
      if (wave < 4) flush(false);
      __syncthreads();
      if (wave >= 4) flush(true);
      __syncthreads();
    }
    NSC_STAMP(4);

    // ---- elementwise epilogue: (channel, 4 consecutive steps) per thread ----
#pragma unroll
    for (int it = 0; it < NEI; ++it) {
      const int id = tid + 512 * it, c = id / (TS / 4), tg = id - c * (TS / 4), t = t0 + 4 * tg;
      if (c < NARROW && t < T) {
        const float* sp = outp + c * OP + 4 * tg;
        f32x4 v = (*reinterpret_cast<const f32x4*>(sp) + *reinterpret_cast<const f32x4*>(sp + NARROW * OP)) +
                  (*reinterpret_cast<const f32x4*>(sp + 2 * NARROW * OP) + *reinterpret_cast<const f32x4*>(sp + 3 * NARROW * OP));
        if (EPI == 2) {
          *reinterpret_cast<f32x4*>(a.out + ((long)b * NARROW + c) * T + t) = v;
        } else if (EPI == 0) {
          v += pv[it];
          f32x4 dl, dg;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float th = ev1[it][i];
            dl[i] = v[i] * th;
            dg[i] = v[i] * ev0[it][i] * (1.f - th * th);
          }
          float* o = a.out + ((long)b * 2 * NARROW + c) * T + t;
          *reinterpret_cast<f32x4*>(o) = dl;
          *reinterpret_cast<f32x4*>(o + (long)NARROW * T) = dg;
        } else {
          f32x4 dz;
#pragma unroll
          for (int i = 0; i < 4; ++i) dz[i] = v[i] * (ev0[it][i] > 0.f ? 1.f : NSC_LRELU_ALPHA);
          *reinterpret_cast<f32x4*>(a.out + ((long)b * NARROW + c) * T + t) = dz;
          if (EPI == 3) *reinterpret_cast<f32x4*>(outp + c * OP + 4 * tg) = dz;   // slot 0 becomes the dz1 tile (this thread's own entry)
        }
      } else if (EPI == 3 && c < NARROW) {
        *reinterpret_cast<f32x4*>(outp + c * OP + 4 * tg) = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (EPI == 3) {
      // ---- dx = (W1^T dz1 + dy) . act'(x): thread = (4 consecutive steps, every 16th channel); dz1 of its steps in registers ----
      __syncthreads();
      constexpr int NTG = TS / 4;                              // 32 groups of 4 steps
      const int tg = tid % NTG, cl0 = tid / NTG;               // 16 channel lanes
      const int t = t0 + 4 * tg;
      const bool ok = t < T;
      f32x4 dz[NARROW];
#pragma unroll
      for (int o = 0; o < NARROW; ++o) dz[o] = *reinterpret_cast<const f32x4*>(outp + o * OP + 4 * tg);
      constexpr int CL = 512 / NTG;                            // 16
      for (int cb = cl0; cb < a.C1; cb += 4 * CL) {            // four channels' loads in flight
        f32x4 yv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = cb + u * CL;
          const long off = ((long)b * a.C1 + (c < a.C1 ? c : cl0)) * T + (ok ? t : 0);
          yv[u] = *reinterpret_cast<const f32x4*>(a.dy + off);
          xv[u] = a.in_act == NSC_ACT_LRELU ? *reinterpret_cast<const f32x4*>(a.x + off) : (f32x4){1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = cb + u * CL;
          if (c < a.C1) {
            const f32x4* w4 = reinterpret_cast<const f32x4*>(w1s + c * NARROW);
            f32x4 acc = yv[u];
#pragma unroll
            for (int o4 = 0; o4 < NARROW / 4; ++o4) {
              const f32x4 wv = w4[o4];
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = fmaf(wv[j], dz[4 * o4 + j][i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] *= xv[u][i] > 0.f ? 1.f : NSC_LRELU_ALPHA;
            if (ok) *reinterpret_cast<f32x4*>(a.dx + ((long)b * a.C1 + c) * T + t) = acc;
          }
        }
      }
      __syncthreads();                                         // slot 0 is rewritten by the next tile's combine
    }
    NSC_STAMP(5);
    // (no barrier here: the next tile's staging writes the planes - last read before the combine barriers - and the slots are
    // rewritten only after the next tile's own staging barrier)
    NSC_STAMP(6);
  }
}

struct Bb1Args {
  const float *dz1, *w1, *dy, *x;
  float* dx;
  int B, C, T, in_act, tpf, chunk;
};
// one input channel (the first block of a decoder stage: x [B,1,T] is broadcast into the residual add): dx[b,0,t] = w1 . dz1 + sum_c dy
__global__ __launch_bounds__(256) void bb_1x1_cin1_kernel(Bb1Args a) {
  __shared__ f32x4 part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / a.tpf, t0 = (blockIdx.x - b * a.tpf) * 256;
  const int T = a.T, t = t0 + 4 * lane;
  const bool ok = t < T;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int c = wave; c < a.C; c += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.dy + ((long)b * a.C + c) * T + (ok ? t : 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += v[i];
  }
  for (int o = wave; o < NARROW; o += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.dz1 + ((long)b * NARROW + o) * T + (ok ? t : 0));
    const float wv = a.w1[o];
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] = fmaf(wv, v[i], s[i]);
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && ok) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (part[0][lane][i] + part[1][lane][i]) + (part[2][lane][i] + part[3][lane][i]);
    *reinterpret_cast<f32x4*>(a.dx + (long)b * T + t) = r;
  }
}

int bb_cu_count() {
  static std::atomic<int> ncu{0};
  int cu = ncu.load(std::memory_order_relaxed);
  if (cu == 0) {
    int dv = 0;
    if (hipGetDevice(&dv) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dv) != hipSuccess) cu = 256;
    ncu.store(cu, std::memory_order_relaxed);
  }
  return cu;
}

template <int CH_, int K_, int DIL_, int NC_, int EPI>
int bb_launch(BbArgs a, hipStream_t st) {
  if (a.src_C == 0) a.src_C = CH_;
  using G = BbGeom<CH_, K_, DIL_, NC_>;
  constexpr size_t smem_k = G::smem + (EPI == 3 ? 100 * NARROW * 4 : 0);       // (+ the 1x1 kernel of up to 100 channels)
  static_assert(smem_k <= 160 * 1024, "LDS of the polyphase data-gradient GEMM");
  auto kern = bb_gemm_kernel<CH_, K_, DIL_, NC_, EPI>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem_k);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "bb_gemm: smem attr: %s", hipGetErrorString(e));
  a.tpf = nsc_cdiv(a.T, G::TS);
  a.ntiles = a.B * a.tpf;
  hipLaunchKernelGGL(kern, dim3(std::min(a.ntiles, bb_cu_count())), dim3(512), smem_k, st, a);
  NSC_CHECK_LAUNCH("bb_gemm");
  return NSC_OK;
}
// the k9 gradient always runs on 50 channels at a time (C = 100: two launches, channels 0..49 | 50..99 - the weight pieces of all 100
// channels, 112 KB, do not fit the LDS next to the planes)
constexpr int bb_w9_words() { return 3 * BbGeom<50, K9, 1, 2>::PLW / 2; }
constexpr int bb_w15_words() { return 3 * BbGeom<2 * NARROW, K15, 1, 2>::PLW / 2; }
static_assert(bb_w9_words() % 4 == 0 && bb_w15_words() % 4 == 0, "every group of pieces starts 16-byte aligned");
}  // namespace

// words of the which = 2 image (nsc_gated_block_simage_words) and its index: per half of 50 output channels the W9 pieces
// [3][9][20][56] (+ 8 zeros per piece), then the Wl | Wr pieces [3][15][20][40] (+ 8 zeros); pairs (o, o + 1) of the [K][20][Cout]
// kernels are neighbours in memory: gather stride 1.  idx arrives filled with -1 (structural zeros: the pad columns of a row and the 8
// zeros behind a piece).
long nsc_bb_simage_words(int C) {
  if (C != 100 && C != 50) return 0;
  return (C / 50) * bb_w9_words() + bb_w15_words();
}
void nsc_bb_simage_index(int C, long w9, long wl, long wr, const int mode_bits[3], int* idx) {
  const int cp9 = 56, plw9 = K9 * NARROW * cp9 + 8, plw15 = K15 * NARROW * 2 * NARROW + 8;
  const int nh = C / 50;
  const long base15 = (long)nh * 3 * plw9 / 2;
  for (int p = 0; p < 3; ++p) {
    for (int hf = 0; hf < nh; ++hf)
      for (int tap = 0; tap < K9; ++tap)
        for (int ci = 0; ci < NARROW; ++ci)
          for (int o = 0; o < 50; o += 2)
            idx[(long)(hf * 3 + p) * plw9 / 2 + ((tap * NARROW + ci) * cp9 + o) / 2] =
                (int)(w9 + ((long)tap * NARROW + ci) * C + 50 * hf + o) | mode_bits[p];
    for (int tap = 0; tap < K15; ++tap)
      for (int ci = 0; ci < NARROW; ++ci)
        for (int c = 0; c < 2 * NARROW; c += 2) {
          const long src = (c < NARROW ? wl : wr) + ((long)tap * NARROW + ci) * NARROW + (c % NARROW);
          idx[base15 + (long)p * plw15 / 2 + ((tap * NARROW + ci) * 2 * NARROW + c) / 2] = (int)src | mode_bits[p];
        }
  }
}

// The data-path backward of one gated block on the which = 2 image: same results as nsc_gated_block_dgrad_img / _simg to fp32 rounding.
// x: the block's input [B,Cin,T] (read only when in_act = lrelu); w1: the 1x1 kernel [Cin][20] in the PARAMETER layout; da must be the
// joint [B,40,T] tensor (dlin | dgate); dx nullable (first block of the network: nobody needs it).
extern "C" int nsc_gated_block_dgrad_simg2(const void* img, const float* w1, const float* x, const float* h, const float* lin, const float* th,
                                           const float* dy, float* dx, float* da, float* dz1, int B, int C, int Cin, int T, int dil,
                                           int in_act, void* stream) {
  NSC_REQUIRE(img && h && lin && th && dy && da && dz1, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_simg2: null pointer");
#ifdef NSC_PROBES
  const bool only_k9 = in_act == -9;                       // (the probes build only: in_act -9 = stop after the k9 launches - tools/dgrad3l_stamps.py)
  if (only_k9) in_act = NSC_ACT_NONE;
#else
  const bool only_k9 = false;
#endif
  NSC_REQUIRE(!dx || (w1 && (x || in_act != NSC_ACT_LRELU)), NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_simg2: dx needs w1 (and x under lrelu)");
  NSC_REQUIRE(B > 0 && T > 0 && (C == 100 || C == 50) && (Cin == C || Cin == 1) && (dil == 1 || dil == 2), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_dgrad_simg2: no kernel for B %d, C %d, Cin %d, T %d, dil %d", B, C, Cin, T, dil);
  NSC_REQUIRE(in_act == NSC_ACT_NONE || in_act == NSC_ACT_LRELU, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_simg2: in_act");
  NSC_REQUIRE(Cin == C || in_act == NSC_ACT_NONE, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_simg2: a one-channel input has no activation");
  NSC_REQUIRE((T & 3) == 0 && (long)B * C * T * 4 < (1L << 31), NSC_ERR_UNSUPPORTED, "nsc_gated_block_dgrad_simg2: needs T %% 4 == 0 and tensors below 2 GB");
  NSC_REQUIRE((((uintptr_t)img | (uintptr_t)h | (uintptr_t)lin | (uintptr_t)th | (uintptr_t)dy | (uintptr_t)da | (uintptr_t)dz1 | (uintptr_t)dx |
                (uintptr_t)x) & 15) == 0, NSC_ERR_UNSUPPORTED, "nsc_gated_block_dgrad_simg2: 16-byte aligned tensors");
  hipStream_t st = (hipStream_t)stream;
  const u16* im = reinterpret_cast<const u16*>(img);
  int rc;
  if (C == 100) {
    // channels 0..49 -> a partial sum parked in dz1 (free until the k15 launch writes it), channels 50..99 + that -> GLU backward
    NSC_REQUIRE(dz1, NSC_ERR_BAD_ARG, "nsc_gated_block_dgrad_simg2: C = 100 needs dz1 (scratch of the first launch)");
    BbArgs a0{dy, im, nullptr, nullptr, nullptr, dz1, B, T, C, 0, 0, 0};
    rc = bb_launch<50, K9, 1, 2, 2>(a0, st);
    if (rc != NSC_OK) return rc;
    BbArgs a1{dy, im + 2 * bb_w9_words(), lin, th, dz1, da, B, T, C, 50, 0, 0};
    rc = bb_launch<50, K9, 1, 2, 0>(a1, st);
  } else {
    BbArgs a9{dy, im, lin, th, nullptr, da, B, T, C, 0, 0, 0};
    rc = bb_launch<50, K9, 1, 2, 0>(a9, st);
  }
  if (rc != NSC_OK || !dz1 || only_k9) return rc;          // (profiling: the first launch(es) alone)
  BbArgs a15{da, im + 2 * (C / 50) * bb_w9_words(), h, nullptr, nullptr, dz1, B, T, 2 * NARROW, 0, 0, 0};
  if (dx && Cin == C) {
    // the 1x1 gradient + residual ride in the k15 launch (EPI 3)
    a15.w1 = w1; a15.dy = dy; a15.x = x; a15.dx = dx; a15.C1 = C; a15.in_act = in_act;
    return dil == 1 ? bb_launch<2 * NARROW, K15, 1, 2, 3>(a15, st) : bb_launch<2 * NARROW, K15, 2, 2, 3>(a15, st);
  }
  rc = dil == 1 ? bb_launch<2 * NARROW, K15, 1, 2, 1>(a15, st) : bb_launch<2 * NARROW, K15, 2, 2, 1>(a15, st);
  if (rc != NSC_OK || !dx) return rc;
  Bb1Args a1{dz1, w1, dy, x, dx, B, C, T, in_act, nsc_cdiv(T, 256), (C + 1) / 2};
  hipLaunchKernelGGL(bb_1x1_cin1_kernel, dim3(B * a1.tpf), dim3(256), 0, st, a1);
  NSC_CHECK_LAUNCH("bb_1x1");
  return NSC_OK;
}
