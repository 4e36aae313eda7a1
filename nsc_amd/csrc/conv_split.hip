// The stride-2 down-sampling conv of the encoder (neural_speech_coding_module.py:229-233: conv1d(h, 100, 9, strides=2) + leaky-relu) and
// its data gradient on the bf16 matrix cores with SPLIT OPERANDS (block_split.hip: every fp32 operand = three bf16 pieces, six
// products, fp32 accumulation: fp32-class results).  The exact-fp32 kernels (conv.hip: conv1d_fwd_m32_kernel, forward and polyphase data
// gradient) run at 77 TF/s on the vector ALUs; these are the four largest launches of a step outside the gated blocks.
//
// Both directions are ONE implicit GEMM  D[m][n] = sum_k Wmat[m][k] X[k][n]  over a [time][100] activation plane with row pitch
// EXACTLY 100, so that k = tap * 100 + c is the flat offset from the column's first row (the Toeplitz-plane idea of block_split.hip):
//   MODE 0 (forward):       m = o (100 -> 7 row tiles), n = output step t, k = (tap 0..8, ci), column t starts at row 2 t - 3 of x
//                           (900 -> 29 k-steps of 32); epilogue + bias, leaky-relu; y [B,100,Tout]
//   MODE 1 (data gradient): the polyphase form (engine._Conv.wtpoly_index): m = 2 ci + p (200 -> 13 row tiles), n = n', k = (t' 0..4, o),
//                           Wmat = W[7 - 2 t' + p][ci][o] (structural zeros where that tap does not exist), column n' starts at row
//                           n' - 2 of dy (500 -> 16 k-steps); epilogue = sub-pixel shuffle dx[ci][2 n' + p]
// A workgroup (8 waves) owns a 64-column tile: it converts the tile's rows of x / dy (fp32 [c][time] in memory) into three bf16 planes
// [time][100] in LDS, then wave (cp = wave & 1, qd = wave >> 1) computes column tiles 2 cp, 2 cp + 1 x its QUARTER of the row tiles over
// the whole K (7 row tiles = 2 | 2 | 2 | 1, 13 = 4 | 3 | 3 | 3: the two waves of a SIMD, w and w + 4, hold 4 | 4 | 3 | 3 resp. 7 | 7 | 6 | 6),
// B fragments from the planes (one 16-byte read per piece, column tile and k-step), A fragments streamed from the kernel-ready IMAGE
// of the weights (16 bytes per lane, piece and (k-step, row tile), each used for two column tiles; the two waves of a quarter read the
// same addresses - one L2 fetch per CU).
// k-slots past the last tap hold zero weights; the rows they read are real (finite) rows of the plane.
#include "nsc_common.h"
#include <algorithm>
#include <type_traits>

#include "block_common.h"
#ifdef NSC_PROBES
extern "C" int nsc_probe_read_conv_split(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsc_dbg_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3;
}
#endif

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#ifndef NSC_EXP
#define NSC_EXP 0
#endif

namespace {
constexpr int CS_C = 100, CS_TT = 64;
template <int MODE>
struct CsGeom {
  static constexpr int STRIDE = MODE == 0 ? 2 : 1, KT = MODE == 0 ? 9 : 5, PADL = MODE == 0 ? 3 : 2;
  static constexpr int M = MODE == 0 ? 100 : 200, NRT = (M + 15) / 16, NRB = NRT / 4, NRREM = NRT % 4;   // quarter qd: NRB + (qd < NRREM) row tiles
  static constexpr int KS = (KT * CS_C + 31) / 32;                       // k-steps of 32
  static constexpr int SHIFT = MODE == 0 ? 1 : 2;                        // the first 16-byte group starts SHIFT samples before row 0
  static constexpr int LASTROW = STRIDE * (CS_TT - 1) + (KS * 32 + CS_C - 1) / CS_C;   // last row any column reads (partly)
  static constexpr int NG = (LASTROW + SHIFT) / 4 + 1;                   // 16-byte groups per channel row
  static constexpr int ROWS = 4 * NG - SHIFT;                            // plane rows (all written)
  static constexpr int PL = ROWS * CS_C;                                 // elements per plane
  static constexpr long IMG_WORDS = (long)KS * NRT * 3 * 256;
};

__device__ __forceinline__ f32x4 cs_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// six products, smallest first (the order of block_split.hip's mfma_split6)
__device__ __forceinline__ f32x4 cs_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
  c = cs_mfma(a[1], b[1], c);
  c = cs_mfma(a[0], b[2], c);
  c = cs_mfma(a[2], b[0], c);
  c = cs_mfma(a[0], b[1], c);
  c = cs_mfma(a[1], b[0], c);
  c = cs_mfma(a[0], b[0], c);
  return c;
}

struct ConvSplitArgs {
  const float* x;        // MODE 0: x [B,100,Tin];  MODE 1: dy [B,100,Tn]
  const uint4* img;      // [k-step][row tile][piece][lane] 16-byte fragments
  const float* bias;     // MODE 0, nullable
  float* y;              // MODE 0: [B,100,Tn];  MODE 1: dx [B,100,2 Tn]
  int B, Tin, Tn;        // Tn = columns per frame (MODE 0: Tout; MODE 1: Tout of the forward conv = length of dy)
  int act, ntiles, tpf;
};

template <int MODE>
__global__ __launch_bounds__(512) void conv_split_kernel(ConvSplitArgs a) {
  using G = CsGeom<MODE>;
  extern __shared__ __attribute__((aligned(16))) u16 cs_sm[];
  // row pitch 200 bytes: MODE 0 columns start at even rows (fragments 16-byte aligned: ds_read_b128), MODE 1 columns at every row
  // (8-byte aligned: two ds_read_b64)
  u16* const plane = cs_sm;
  constexpr int PLS = (G::PL + 8 + 7) & ~7;                              // plane stride (elements), 16-byte multiple
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int cp2 = wave & 1, qd = wave >> 1;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tpf, t0 = (tile - b * a.tpf) * CS_TT;
    const float* xb = a.x + (long)b * CS_C * a.Tin;
    NSC_STAMP(0);
    // ---- stage: rows u = STRIDE t0 - PADL + r (r = 0 .. ROWS-1) of all 100 channels -> three bf16 planes [r][ch] ----
    // a unit = (channel pair, 16-byte group of 4 samples); a wave takes 8 pairs x 8 groups: 128 contiguous bytes per channel row from
    // memory, and LDS words that are at most 2-way bank conflicts
    {
      const int u_al = G::STRIDE * t0 - G::PADL - G::SHIFT;               // multiple of 4
      constexpr int NCPB = (CS_C / 2 + 7) / 8, NGB = (G::NG + 7) / 8;     // blocks of 8 pairs / 8 groups
      constexpr int NIT = (NCPB * NGB + 7) / 8;                           // units per wave
      const int gl = lane & 7, cl = lane >> 3;
      f32x4 v0[NIT], v1[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {                                  // all loads of the tile in flight at once
        const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
        const int cp = cpb * 8 + cl, g = gb * 8 + gl, u = u_al + 4 * g;
        v0[it] = v1[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ub < NCPB * NGB && cp < CS_C / 2 && g < G::NG && u >= 0 && u < a.Tin) {
          v0[it] = *reinterpret_cast<const f32x4*>(xb + (long)(2 * cp) * a.Tin + u);
          v1[it] = *reinterpret_cast<const f32x4*>(xb + (long)(2 * cp + 1) * a.Tin + u);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
        const int cp = cpb * 8 + cl, g = gb * 8 + gl;
        if (ub < NCPB * NGB && cp < CS_C / 2 && g < G::NG && !((NSC_EXP & 256) && tile != (int)blockIdx.x)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j - G::SHIFT;
            if (r >= 0) {
              unsigned pk[3];
              nsc_split2(v0[it][j], v1[it][j], pk);
              unsigned* w = reinterpret_cast<unsigned*>(plane + r * CS_C + 2 * cp);
#pragma unroll
              for (int p = 0; p < 3; ++p) w[p * (PLS / 2)] = pk[p];
            }
          }
        }
      }
    }
    NSC_STAMP(1);
    __syncthreads();
    NSC_STAMP(2);

    // ---- the GEMM: wave (cp2, qd): column tiles 2 cp2 + {0, 1}, row tiles [rt0, rt0 + NR) ----
    const u16* bcol = plane + (G::STRIDE * (cp2 * 32 + l15)) * CS_C + 8 * q;      // second column tile: + STRIDE * 16 rows
    const uint4* imgl = a.img + lane;
    auto run = [&](auto nr_c, int rt0) {
      constexpr int NR = decltype(nr_c)::value;
      f32x4 acc[2][NR];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[e][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 abuf[2][NR][3];
      auto load_a = [&](int s, bf16x8 (&dst)[NR][3]) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int p = 0; p < 3; ++p) dst[r][p] = __builtin_bit_cast(bf16x8, imgl[((long)(s * G::NRT + rt0 + r) * 3 + p) * 64]);
      };
      auto load_b = [&](int s, int e, bf16x8 (&dst)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u16* ptr = bcol + e * (G::STRIDE * 16 * CS_C) + p * PLS + 32 * s;
          if (MODE == 0) {
            dst[p] = *reinterpret_cast<const bf16x8*>(ptr);
          } else {
            const uint2 lo = *reinterpret_cast<const uint2*>(ptr), hi = *reinterpret_cast<const uint2*>(ptr + 4);
            const i32x4_t v = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
            dst[p] = __builtin_bit_cast(bf16x8, v);
          }
        }
      };
      // timing experiments (make exp EXP=64|128|256, WRONG RESULTS): 64 = the A fragments are loop-invariant (no image stream),
      // 128 = the B fragments are (no LDS operand reads), 256 = no staging
      // One k-step: the NEXT step's fragments are requested first (A from the image: an L2 round trip; B from the planes), a scheduling
      // fence keeps the requests up there (left alone, hipcc re-used the idle A buffer's registers for this step's B fragments and sank
      // the image loads to the END of the step: every other step waited out a full L2 latency), then the 12 NR products - product-major,
      // so that consecutive MFMAs go to different accumulators (a dependent bf16 MFMA issues ~12 cycles late).
      bf16x8 bbuf[2][2][3];
      auto step = [&](int s, bf16x8 (&cur)[NR][3], bf16x8 (&nxt)[NR][3], bf16x8 (&bc)[2][3], bf16x8 (&bn)[2][3]) {
        {   // unconditional (the last step re-requests its own fragments): a branch here makes hipcc's vmcnt bookkeeping assume the
            // shorter path and wait for the NEW requests before the first MFMA
          const int sn = s + 1 < G::KS ? s + 1 : G::KS - 1;
          if (!(NSC_EXP & 64)) load_a(sn, nxt);
          load_b((NSC_EXP & 128) ? 0 : sn, 0, bn[0]);
          load_b((NSC_EXP & 128) ? 0 : sn, 1, bn[1]);
        }
        if (NSC_EXP & 64) {
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int p = 0; p < 3; ++p) nxt[r][p] = cur[r][p];
        }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PW[6] = {1, 0, 2, 0, 1, 0}, PX[6] = {1, 2, 0, 1, 0, 0};      // (weight piece, activation piece), smallest first
#pragma unroll
        for (int pi = 0; pi < 6; ++pi)
#pragma unroll
          for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              if (MODE == 0) acc[e][r] = cs_mfma(bc[e][PX[pi]], cur[r][PW[pi]], acc[e][r]);   // transposed product: a lane's 4 values = 4 consecutive steps of one o
              else acc[e][r] = cs_mfma(cur[r][PW[pi]], bc[e][PX[pi]], acc[e][r]);
            }
        __builtin_amdgcn_sched_barrier(0);
      };
      load_a(0, abuf[0]);
      load_b(0, 0, bbuf[0][0]);
      load_b(0, 1, bbuf[0][1]);
#pragma unroll 1
      for (int s = 0; s + 1 < G::KS; s += 2) {
        step(s, abuf[0], abuf[1], bbuf[0], bbuf[1]);
        step(s + 1, abuf[1], abuf[0], bbuf[1], bbuf[0]);
      }
      if (G::KS & 1) step(G::KS - 1, abuf[0], abuf[1], bbuf[0], bbuf[1]);
      NSC_STAMP(3);
      // ---- epilogue ----
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int ct = 2 * cp2 + e;
        if (MODE == 0) {
          // D^T: acc[r][i] = y[o = (rt0 + r) 16 + l15][t0 + 16 ct + 4 q + i]
          float* yb = a.y + (long)b * CS_C * a.Tn + t0 + 16 * ct + 4 * q;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int o = (rt0 + r) * 16 + l15;
            if (o < CS_C) {
              const float bv = a.bias ? a.bias[o] : 0.f;
              f32x4 v = acc[e][r];
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] = nsc_apply_act(v[i] + bv, a.act);
              *reinterpret_cast<f32x4*>(yb + (long)o * a.Tn) = v;
            }
          }
        } else {
          // D: acc[r][i] = row m = (rt0 + r) 16 + 4 q + i = 2 ci + p, column n' = t0 + 16 ct + l15  ->  dx[ci][2 n' + p]
          float* yb = a.y + (long)b * CS_C * 2 * a.Tn + 2 * (t0 + 16 * ct + l15);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int ci = ((rt0 + r) * 16 + 4 * q) >> 1;
            if (ci < CS_C) {
              typedef float f32x2_ __attribute__((ext_vector_type(2)));
              *reinterpret_cast<f32x2_*>(yb + (long)ci * 2 * a.Tn) = (f32x2_){acc[e][r][0], acc[e][r][1]};
              *reinterpret_cast<f32x2_*>(yb + (long)(ci + 1) * 2 * a.Tn) = (f32x2_){acc[e][r][2], acc[e][r][3]};
            }
          }
        }
      }
    };
    {
      const int rt0 = qd * G::NRB + (qd < G::NRREM ? qd : G::NRREM);
      if (qd < G::NRREM) run(std::integral_constant<int, G::NRB + 1>{}, rt0);
      else run(std::integral_constant<int, G::NRB>{}, rt0);
    }
    NSC_STAMP(4);
    __syncthreads();                                      // the planes are rewritten by the next tile
    NSC_STAMP(5);
  }
}

template <int MODE>
size_t cs_smem() {
  using G = CsGeom<MODE>;
  const int pls = (G::PL + 8 + 7) & ~7;
  return (size_t)(3 * pls + 8) * sizeof(u16);
}

bool cs_shape_ok(const nsc_conv_desc* d) {
  return d->Cin == CS_C && d->Cout == CS_C && d->K == 9 && d->stride == 2 && d->dil == 1 && d->padL == 3 && d->in_up == 0 &&
         d->Tout * 2 == d->Tin && d->Tout % CS_TT == 0;
}
int cs_mode(int plane, int stride) {                       // misc.hip: gather_word
  const int sel = stride == 20 ? 0 : (stride == 25 ? 1 : (stride == 50 ? 2 : (stride == 100 ? 3 : 4)));
  return (1 + plane * 5 + sel) << 26;
}
}  // namespace

extern "C" long nsc_conv1d_simage_words(int which, const nsc_conv_desc* d) {
  if (!d || !cs_shape_ok(d) || which < 0 || which > 1) return 0;
  return which == 0 ? CsGeom<0>::IMG_WORDS : CsGeom<1>::IMG_WORDS;
}

extern "C" int nsc_conv1d_simage_index(int which, const nsc_conv_desc* d, long w_off, int* idx) {
  NSC_REQUIRE(d && idx && (which == 0 || which == 1), NSC_ERR_BAD_ARG, "nsc_conv1d_simage_index: bad args");
  NSC_REQUIRE(cs_shape_ok(d), NSC_ERR_UNSUPPORTED, "nsc_conv1d_simage_index: the split-operand conv serves the stride-2 k9 100 -> 100 conv only");
  NSC_REQUIRE(w_off >= 0 && w_off + 9L * CS_C * CS_C < (1L << 26), NSC_ERR_BAD_ARG, "nsc_conv1d_simage_index: offset out of range");
  const int ks = which == 0 ? CsGeom<0>::KS : CsGeom<1>::KS, nrt = which == 0 ? CsGeom<0>::NRT : CsGeom<1>::NRT;
  const int M = which == 0 ? 100 : 200, kt = which == 0 ? 9 : 5;
  for (int s = 0; s < ks; ++s)
    for (int rt = 0; rt < nrt; ++rt)
      for (int p = 0; p < 3; ++p)
        for (int lane = 0; lane < 64; ++lane)
          for (int jw = 0; jw < 4; ++jw) {
            const int m = rt * 16 + (lane & 15), k = 32 * s + 8 * (lane >> 4) + 2 * jw, tp = k / CS_C, cc = k - tp * CS_C;
            int v = -1;
            if (m < M && tp < kt) {
              if (which == 0) {
                v = (int)(w_off + ((long)tp * CS_C + cc) * CS_C + m) | cs_mode(p, CS_C);                 // W[tap][ci][o]; pair ci, ci + 1
              } else {
                const int kk = 7 - 2 * tp + (m & 1);
                if (kk >= 0 && kk < 9) v = (int)(w_off + ((long)kk * CS_C + (m >> 1)) * CS_C + cc) | cs_mode(p, 1);   // pair o, o + 1
              }
            }
            idx[((((long)s * nrt + rt) * 3 + p) * 64 + lane) * 4 + jw] = v;
          }
  return NSC_OK;
}

template <int MODE>
static int cs_launch(const ConvSplitArgs& a, hipStream_t st) {
  auto kern = conv_split_kernel<MODE>;
  const size_t smem = cs_smem<MODE>();
  const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv_split: smem attr: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(std::min(a.ntiles, 256)), dim3(512), smem, st, a);
  NSC_CHECK_LAUNCH("conv_split");
  return NSC_OK;
}

extern "C" int nsc_conv1d_fwd_simg(const nsc_conv_desc* d, const float* x, const void* image, const float* bias, float* y, void* stream) {
  NSC_REQUIRE(d && x && image && y, NSC_ERR_BAD_ARG, "nsc_conv1d_fwd_simg: null pointer");
  NSC_REQUIRE(cs_shape_ok(d) && d->res_mode == 0 && d->mul_mode == 0 && d->out_mode == 0 && !d->accumulate &&
                  (d->act == NSC_ACT_NONE || d->act == NSC_ACT_LRELU || d->act == NSC_ACT_TANH),
              NSC_ERR_UNSUPPORTED, "nsc_conv1d_fwd_simg: the stride-2 k9 100 -> 100 conv (Tout %% 64 == 0, plain epilogue) only");
  NSC_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)image) & 15) == 0, NSC_ERR_UNSUPPORTED, "nsc_conv1d_fwd_simg: 16-byte aligned tensors");
  ConvSplitArgs a{x, (const uint4*)image, bias, y, d->B, d->Tin, d->Tout, d->act, d->B * (d->Tout / CS_TT), d->Tout / CS_TT};
  return cs_launch<0>(a, (hipStream_t)stream);
}

extern "C" int nsc_conv1d_dgrad_simg(const nsc_conv_desc* d, const float* dy, const void* image, float* dx, void* stream) {
  NSC_REQUIRE(d && dy && image && dx, NSC_ERR_BAD_ARG, "nsc_conv1d_dgrad_simg: null pointer");
  NSC_REQUIRE(cs_shape_ok(d), NSC_ERR_UNSUPPORTED, "nsc_conv1d_dgrad_simg: the stride-2 k9 100 -> 100 conv (Tout %% 64 == 0) only");
  NSC_REQUIRE((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)image) & 15) == 0, NSC_ERR_UNSUPPORTED, "nsc_conv1d_dgrad_simg: 16-byte aligned tensors");
  ConvSplitArgs a{dy, (const uint4*)image, nullptr, dx, d->B, d->Tout, d->Tout, 0, d->B * (d->Tout / CS_TT), d->Tout / CS_TT};
  return cs_launch<1>(a, (hipStream_t)stream);
}
