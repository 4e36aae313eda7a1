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

    // ---- the GEMM: a wave computes row tiles [rt0, rt0 + NR) x column tiles [ct0, ct0 + NC) ----
    const uint4* imgl = a.img + lane;
    auto run = [&](auto nr_c, auto nc_c, int rt0, int ct0) {
      constexpr int NR = decltype(nr_c)::value, NC = decltype(nc_c)::value;
      const u16* bcol = plane + (G::STRIDE * (ct0 * 16 + l15)) * CS_C + 8 * q;      // next column tile: + STRIDE * 16 rows
      f32x4 acc[NC][NR];
#pragma unroll
      for (int e = 0; e < NC; ++e)
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
      bf16x8 bbuf[2][NC][3];
      auto step = [&](int s, bf16x8 (&cur)[NR][3], bf16x8 (&nxt)[NR][3], bf16x8 (&bc)[NC][3], bf16x8 (&bn)[NC][3]) {
        {   // unconditional (the last step re-requests its own fragments): a branch here makes hipcc's vmcnt bookkeeping assume the
            // shorter path and wait for the NEW requests before the first MFMA
          const int sn = s + 1 < G::KS ? s + 1 : G::KS - 1;
          if (!(NSC_EXP & 64)) load_a(sn, nxt);
#pragma unroll
          for (int e = 0; e < NC; ++e) load_b((NSC_EXP & 128) ? 0 : sn, e, bn[e]);
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
            for (int e = 0; e < NC; ++e) {
              if (MODE == 0) acc[e][r] = cs_mfma(bc[e][PX[pi]], cur[r][PW[pi]], acc[e][r]);   // transposed product: a lane's 4 values = 4 consecutive steps of one o
              else acc[e][r] = cs_mfma(cur[r][PW[pi]], bc[e][PX[pi]], acc[e][r]);
            }
        __builtin_amdgcn_sched_barrier(0);
      };
      load_a(0, abuf[0]);
#pragma unroll
      for (int e = 0; e < NC; ++e) load_b(0, e, bbuf[0][e]);
#pragma unroll 1
      for (int s = 0; s + 1 < G::KS; s += 2) {
        step(s, abuf[0], abuf[1], bbuf[0], bbuf[1]);
        step(s + 1, abuf[1], abuf[0], bbuf[1], bbuf[0]);
      }
      if (G::KS & 1) step(G::KS - 1, abuf[0], abuf[1], bbuf[0], bbuf[1]);
      NSC_STAMP(3);
      // ---- epilogue ----
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        const int ct = ct0 + e;
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
#ifndef NSC_CS_MAP
#define NSC_CS_MAP 0
#endif
    if (NSC_CS_MAP == 1 && MODE == 0) {
      // A/B: one row tile x all four column tiles per wave (wave 7 idle): every image fragment fetched once per workgroup
      if (wave < 7) run(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, wave, 0);
    } else if (NSC_CS_MAP == 1 && MODE == 1) {
      if (wave < 5) run(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{}, 2 * wave, 0);
      else run(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, 10 + (wave - 5), 0);
    } else {
      const int rt0 = qd * G::NRB + (qd < G::NRREM ? qd : G::NRREM);
      if (qd < G::NRREM) run(std::integral_constant<int, G::NRB + 1>{}, std::integral_constant<int, 2>{}, rt0, 2 * cp2);
      else run(std::integral_constant<int, G::NRB>{}, std::integral_constant<int, 2>{}, rt0, 2 * cp2);
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

// =====================================================================================================
// WEIGHT GRADIENT of the same conv on split operands:  dW[k][ci][o] += sum_{b,t} x[b][ci][2 t + k - 3] dy[b][o][t],  db[o] += sum dy.
// GEMM rows m = (k, ci) (900 -> 57 row tiles), columns o (100 -> 7 column tiles), reduction over (b, t).  The 399 accumulator tiles are
// 408 KB - most of a CU's register file - so a job is split into two PARTS (row tiles 0..28 | 29..56), each served by its own
// persistent workgroups: wave w keeps row tiles w, w + 8, w + 16, w + 24 of its part x all 7 column tiles (112 registers) and walks
// 64-step tiles of (frame, time).  Per tile: x is staged exactly as in the forward kernel (three bf16 planes [2 t0 - 3 + r][100], row
// pitch exactly 100, so row m of the virtual im2col matrix at output step tl is element 2 tl * 100 + m); the A fragment wants 8
// consecutive STEPS of one row - the transpose of what is contiguous - and comes from two ds_read_b64_tr_b16 (block_split.hip, weight
// gradients); dy is staged as [o][t] planes (row pitch 72) and read as 16-byte fragments.  Both operands come from LDS, so the NEXT
// tile's x and dy travel in registers during the MFMAs.  A workgroup ends by storing its accumulators to a private slab; one reduce
// launch sums the slabs into dW / db (accumulating).
// =====================================================================================================
namespace {
constexpr int CW_LDT = 72, CW_NRT = 57, CW_NRT0 = 29, CW_NCT = 7, CW_MAXJ = 8;
constexpr int CW_ROWS0 = CW_NRT0 * 16;                      // 464 rows of (k, ci) in part 0
constexpr int CW_DB_OFF = CW_ROWS0 * CS_C;                  // db behind part 0's rows in its slabs
constexpr long CW_SLAB = ((long)CW_DB_OFF + CS_C + 63) & ~63L;
struct ConvWgradSplitJob { const float* x; const float* dy; float* dw; float* db; int Tin, Tout, ntiles, tpf, wg0, nwg; };
struct ConvWgradSplitBatch { ConvWgradSplitJob j[CW_MAXJ]; int njobs, B; float* slabs; };

typedef short s16x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 cw_frag_tr(const u16* p0, const u16* p1) {
  typedef __attribute__((address_space(3))) s16x4_* lds_p;
  const s16x4_ v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4_ v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  return (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}

__global__ __launch_bounds__(512) void conv_wgrad_split_kernel(ConvWgradSplitBatch t) {
  using G = CsGeom<0>;
  extern __shared__ __attribute__((aligned(16))) u16 cs_sm[];
  constexpr int PLS = (G::PL + 8 + 7) & ~7;
  constexpr int PLD = CW_NCT * 16 * CW_LDT;
  u16* const xpl = cs_sm;
  u16* const dpl = cs_sm + 3 * PLS;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4, tq = l15 >> 2, tp = l15 & 3;
  int ji = 0;
  for (int q = 1; q < t.njobs; ++q)
    if ((int)blockIdx.x >= t.j[q].wg0) ji = q;
  const ConvWgradSplitJob& jb = t.j[ji];
  const int wl = blockIdx.x - jb.wg0, part = wl / jb.nwg, slot = wl - part * jb.nwg;
  const int rt0 = (part ? CW_NRT0 : 0) + wave;                          // this wave's row tiles: rt0 + 8 r
  const bool has4 = rt0 + 24 < (part ? CW_NRT : CW_NRT0);

  f32x4 acc[4][CW_NCT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < CW_NCT; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging: units of the x tile as in conv_split_kernel<0>; units (o, 16-byte group) of the dy tile ----
  constexpr int NCPB = (CS_C / 2 + 7) / 8, NGB = (G::NG + 7) / 8, NITX = (NCPB * NGB + 7) / 8;
  constexpr int NITD = (CS_C * (CS_TT / 4) + 511) / 512;
  const int gl = lane & 7, cl = lane >> 3;
  f32x4 xv0[NITX], xv1[NITX], dv[NITD];
  float bs[NITD];
#pragma unroll
  for (int it = 0; it < NITD; ++it) bs[it] = 0.f;
  auto load_tile = [&](int tile) {
    const int b = tile / jb.tpf, t0 = (tile - b * jb.tpf) * CS_TT;
    const float* xb = jb.x + (long)b * CS_C * jb.Tin;
    const int u_al = 2 * t0 - G::PADL - G::SHIFT;
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
      const int cp = cpb * 8 + cl, g = gb * 8 + gl, u = u_al + 4 * g;
      xv0[it] = xv1[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ub < NCPB * NGB && cp < CS_C / 2 && g < G::NG && u >= 0 && u < jb.Tin) {
        xv0[it] = *reinterpret_cast<const f32x4*>(xb + (long)(2 * cp) * jb.Tin + u);
        xv1[it] = *reinterpret_cast<const f32x4*>(xb + (long)(2 * cp + 1) * jb.Tin + u);
      }
    }
    const float* db_ = jb.dy + (long)b * CS_C * jb.Tout + t0;
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int id = it * 512 + tid, o = id >> 4, g = id & 15;
      dv[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (o < CS_C) dv[it] = *reinterpret_cast<const f32x4*>(db_ + (long)o * jb.Tout + 4 * g);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < NITX; ++it) {
      const int ub = wave + 8 * it, cpb = ub / NGB, gb = ub - cpb * NGB;
      const int cp = cpb * 8 + cl, g = gb * 8 + gl;
      if (ub < NCPB * NGB && cp < CS_C / 2 && g < G::NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j - G::SHIFT;
          if (r >= 0) {
            unsigned pk[3];
            nsc_split2(xv0[it][j], xv1[it][j], pk);
            unsigned* w = reinterpret_cast<unsigned*>(xpl + r * CS_C + 2 * cp);
#pragma unroll
            for (int p = 0; p < 3; ++p) w[p * (PLS / 2)] = pk[p];
          }
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      const int id = it * 512 + tid, o = id >> 4, g = id & 15;
      if (o < CS_C) {
        unsigned pa[3], pb[3];
        nsc_split2(dv[it][0], dv[it][1], pa);
        nsc_split2(dv[it][2], dv[it][3], pb);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dpl + p * PLD + o * CW_LDT + 4 * g) = make_uint2(pa[p], pb[p]);
        bs[it] += (dv[it][0] + dv[it][1]) + (dv[it][2] + dv[it][3]);
      }
    }
  };

  // fragment addresses: A = transposed reads of the x planes (block row tq = step 32 s + 8 kq + tq (+ 4), columns 16 rt + 4 tp ..),
  // B = [o][t] planes, row 16 ct + l15, k = 32 s + 8 kq ..
  const u16* const atr = xpl + (2 * (8 * kq + tq)) * CS_C + 16 * rt0 + 4 * tp;
  const u16* const bfr = dpl + l15 * CW_LDT + 8 * kq;

  if (slot < jb.ntiles) load_tile(slot);
  for (int tile = slot; tile < jb.ntiles; tile += jb.nwg) {
    __syncthreads();                                   // everyone is done reading the previous tile
    store_tile();
    __syncthreads();
    if (tile + jb.nwg < jb.ntiles) load_tile(tile + jb.nwg);       // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int rp = 0; rp < 2; ++rp) {
        if (rp == 1 && !has4) {
          // (waves with three row tiles: the pair's second tile is skipped below)
        }
        bf16x8 af[2][3];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u16* q0 = atr + p * PLS + s * (64 * CS_C) + (2 * rp + r2) * 128;
            af[r2][p] = cw_frag_tr(q0, q0 + 8 * CS_C);
          }
#pragma unroll
        for (int c = 0; c < CW_NCT; ++c) {
          bf16x8 bf[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const bf16x8*>(bfr + p * PLD + c * 16 * CW_LDT + 32 * s);
          acc[2 * rp][c] = cs_split6(af[0], bf, acc[2 * rp][c]);
          if (rp == 0 || has4) acc[2 * rp + 1][c] = cs_split6(af[1], bf, acc[2 * rp + 1][c]);
          if (c & 1) __builtin_amdgcn_sched_barrier(0);       // (keeps hipcc from hoisting all 21 B fragments: registers)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- flush to this workgroup's slab ----
  float* slab = t.slabs + (long)blockIdx.x * CW_SLAB;
  const int m_base = part ? CW_ROWS0 : 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (r == 3 && !has4) break;
#pragma unroll
    for (int c = 0; c < CW_NCT; ++c) {
      const int o = 16 * c + l15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * (rt0 + 8 * r) + 4 * kq + i;
        if (o < CS_C && m < 9 * CS_C) slab[(long)(m - m_base) * CS_C + o] = acc[r][c][i];
      }
    }
  }
  if (part == 0) {
#pragma unroll
    for (int it = 0; it < NITD; ++it) {
      float v = bs[it];
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
      const int o = (it * 512 + tid) >> 4;
      if ((tid & 15) == 0 && o < CS_C) slab[CW_DB_OFF + o] = v;
    }
  }
}

// dw / db += sum of the part's slabs; blockIdx.y = job * 2 + part; a workgroup owns 64 elements, its four waves take every fourth slab
__global__ __launch_bounds__(256) void conv_wgrad_split_reduce_kernel(ConvWgradSplitBatch t) {
  __shared__ float part_[4][64];
  const int ji = blockIdx.y >> 1, part = blockIdx.y & 1;
  const ConvWgradSplitJob& jb = t.j[ji];
  const int n = part ? (9 * CS_C - CW_ROWS0) * CS_C : CW_DB_OFF + CS_C;
  const float* s0 = t.slabs + (long)(jb.wg0 + part * jb.nwg) * CW_SLAB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = blockIdx.x * 64; base < n; base += gridDim.x * 64) {
    const int i = base + lane;
    float a0 = 0.f, a1 = 0.f;
    if (i < n) {
      int w = wave;
      for (; w + 4 < jb.nwg; w += 8) {
        a0 += s0[(long)w * CW_SLAB + i];
        a1 += s0[(long)(w + 4) * CW_SLAB + i];
      }
      for (; w < jb.nwg; w += 4) a0 += s0[(long)w * CW_SLAB + i];
    }
    part_[wave][lane] = a0 + a1;
    __syncthreads();
    if (wave == 0 && i < n) {
      const float v = (part_[0][lane] + part_[1][lane]) + (part_[2][lane] + part_[3][lane]);
      if (part == 0 && i >= CW_DB_OFF) { if (jb.db) jb.db[i - CW_DB_OFF] += v; }
      else jb.dw[(long)(part ? CW_ROWS0 : 0) * CS_C + i] += v;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" long nsc_conv1d_wgrad_split_workspace() { return 256L * CW_SLAB; }

// jobs: nsc_conv_wgrad_job (x = the conv's input, dz = dy, dw / db accumulated into, flip_taps 0) of stride-2 k9 100 -> 100 convs
extern "C" int nsc_conv1d_wgrad_split(const nsc_conv_wgrad_job* jobs, int njobs, float* workspace, long workspace_floats, void* stream) {
  NSC_REQUIRE(jobs && njobs > 0 && workspace, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_split: bad arguments");
  NSC_REQUIRE(njobs <= CW_MAXJ, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_split: more than %d jobs", CW_MAXJ);
  NSC_REQUIRE(workspace_floats >= 256L * CW_SLAB && ((uintptr_t)workspace & 15) == 0, NSC_ERR_BAD_ARG,
              "nsc_conv1d_wgrad_split: workspace %ld floats < %ld", workspace_floats, 256L * CW_SLAB);
  ConvWgradSplitBatch t;
  memset(&t, 0, sizeof(t));
  t.njobs = njobs;
  t.slabs = workspace;
  const int per = 256 / (2 * njobs);                       // workgroups per (job, part)
  int wg = 0;
  for (int q = 0; q < njobs; ++q) {
    const nsc_conv_wgrad_job& jb = jobs[q];
    NSC_REQUIRE(jb.x && jb.dz && jb.dw, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_split: job %d: null x/dz/dw", q);
    NSC_REQUIRE(cs_shape_ok(&jb.d) && !jb.flip_taps, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_split: job %d: the stride-2 k9 100 -> 100 conv only", q);
    NSC_REQUIRE((((uintptr_t)jb.x | (uintptr_t)jb.dz) & 15) == 0, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_split: 16-byte aligned tensors");
    ConvWgradSplitJob& j = t.j[q];
    j.x = jb.x; j.dy = jb.dz; j.dw = jb.dw; j.db = jb.db; j.Tin = jb.d.Tin; j.Tout = jb.d.Tout;
    j.tpf = jb.d.Tout / CS_TT; j.ntiles = jb.d.B * j.tpf;
    j.nwg = std::max(1, std::min(per, j.ntiles)); j.wg0 = wg;
    wg += 2 * j.nwg;
  }
  hipStream_t st = (hipStream_t)stream;
  auto kern = conv_wgrad_split_kernel;
  using G = CsGeom<0>;
  const size_t smem = (size_t)(3 * ((G::PL + 8 + 7) & ~7) + 3 * CW_NCT * 16 * CW_LDT) * sizeof(u16);
  const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv_wgrad_split: smem attr: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(wg), dim3(512), smem, st, t);
  NSC_CHECK_LAUNCH("conv_wgrad_split");
  hipLaunchKernelGGL(conv_wgrad_split_reduce_kernel, dim3(256, 2 * njobs), dim3(256), 0, st, t);
  NSC_CHECK_LAUNCH("conv_wgrad_split_reduce");
  return NSC_OK;
}
