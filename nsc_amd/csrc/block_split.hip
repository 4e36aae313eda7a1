// block_split.hip - the gated bottleneck block (nn_core_operator.py:82-112) with its two long contractions on the bf16 MATRIX CORES,
// fp32 operands SPLIT into three bf16 pieces, six products, fp32 accumulation ("3 x bf16": 24 significand bits, the error class of the
// fp32 instruction - tools/mfma_bf16_split.hip, profiles/r04d_mfma_bf16_split.txt).
//
// Why: on gfx950 the fp32 matrix instruction (v_mfma_f32_16x16x4_f32, 32 cycles, 1024 MACs) IS the vector ALU: 157 TF/s, and every
// staging / gate / copy-out instruction of the partner wave waits for it.  v_mfma_f32_16x16x32_bf16 does 8192 MACs in
// 16 cycles on the matrix cores and holds the vector issue for 8 of them: six products per fp32 product are still 2.7x the rate, and
// the elementwise work of the other wave of a SIMD proceeds underneath.
//
//   x = hi + lo + lo2 :  hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo)     (round to nearest even; exact: 3 x 8 bits)
//   a . b  ~  al.bl + ah.bl2 + al2.bh + ah.bl + al.bh + ah.bh                          (dropped terms <= 2^-24 |a||b|)
//
// LAYOUT.  A bf16 MFMA fragment is 8 CONSECUTIVE k of one row / column per lane, so the reduction index must be contiguous in LDS.
// The 20-channel intermediates h and g live as [time][20] bf16 planes with a row pitch of EXACTLY 20 elements: the reduction index of
// a k-tap conv, k = tap * 20 + ci, is then the flat offset from the output column's own row - (t + tap) * 20 + ci = t * 20 + k - and a
// B fragment is 16 contiguous bytes at (column * 20 + 32 s + 8 q) elements (8-byte aligned: two ds_read_b64).  No im2col, no per-tap
// addressing, 300 of 320 (k15) and 180 of 192 (k9) k-slots useful.  Dilation 2 keeps the two time parities in separate halves of the
// plane (a tap moves time by 2, so an output column only ever reads its own parity): column j -> (j & 1) * HALF + (j >> 1) * 20.
// The k-slots past the last tap read into the following rows: the WEIGHTS are zero there, and the rows hold finite data by
// construction (every column of a plane is written - zeros outside the frame - before it is read).
//
// This file: the forward kernel (phase 1, the 1x1 C -> 20, stays on the exact fp32 instruction with x staged as fp32 rows - 6 % of the
// block's MACs, and the residual wants x in fp32 anyway; phases 2 and 3 - both k15 gate convs and the k9 conv, 94 % - run split).
// Same tiling, chains, prefetch and pair-launch protocol as gated_block_fwd2_body (block.hip): read that header first.
#include "nsc_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "block_args.h"
#include "block_common.h"

#ifdef NSC_PROBES
extern "C" int nsc_probe_read_split(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsc_dbg_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -3;
}
#endif

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
#ifndef NSC_EXP
#define NSC_EXP 0   // timing experiments (make exp EXP=n, WRONG RESULTS): 512 = phase-2 B fragments loop-invariant, 1024 = phase-2 A fragments, 2048 = phase-3 B fragments
#endif

__device__ __forceinline__ f32x4 mfma_bf(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// the six products of one k-step, smallest first (NSC_SPLIT_ORDER = 1: largest first - an A/B switch of the build, same error class)
#ifndef NSC_SPLIT_ORDER
#define NSC_SPLIT_ORDER 0
#endif
__device__ __forceinline__ f32x4 mfma_split6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x4 c) {
#if NSC_SPLIT_ORDER == 0
  c = mfma_bf(a[1], b[1], c);
  c = mfma_bf(a[0], b[2], c);
  c = mfma_bf(a[2], b[0], c);
  c = mfma_bf(a[0], b[1], c);
  c = mfma_bf(a[1], b[0], c);
  c = mfma_bf(a[0], b[0], c);
#else
  c = mfma_bf(a[0], b[0], c);
  c = mfma_bf(a[1], b[0], c);
  c = mfma_bf(a[0], b[1], c);
  c = mfma_bf(a[2], b[0], c);
  c = mfma_bf(a[0], b[2], c);
  c = mfma_bf(a[1], b[1], c);
#endif
  return c;
}
// a fragment (8 bf16) from LDS: 8-byte aligned (activation planes: ds_read2_b64) / 16-byte aligned (weight images: ds_read_b128)
__device__ __forceinline__ bf16x8 ld_frag8(const u16* p) {
  const uint2 a = *reinterpret_cast<const uint2*>(p), b = *reinterpret_cast<const uint2*>(p + 4);
  const i32x4_t v = {(int)a.x, (int)a.y, (int)b.x, (int)b.y};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 ld_frag16(const u16* p) { return *reinterpret_cast<const bf16x8*>(p); }
// An LDS address the compiler cannot see through: constant offsets added AFTER this point stay in the offset field of the ds
// instruction (ds_read2_b64 has 8-bit offsets in units of 8 bytes; with the region's base visible, hipcc folds base + offset into
// one constant too large for the field and keeps one address register per access - 30 of them in phase 2).
typedef __attribute__((address_space(3))) const u16* nsc_lds_cu16;
__device__ __forceinline__ nsc_lds_cu16 nsc_opaque_lds(const u16* p) {
  unsigned a = (unsigned)(unsigned long long)(nsc_lds_cu16)p;
  asm volatile("" : "+v"(a));
  return (nsc_lds_cu16)(unsigned long long)a;
}
// NSC_FRAG8_SPLIT (A/B build switch, `make EXTRA=-DNSC_FRAG8_SPLIT=1`): the second half through an address hipcc cannot see next to
// the first, so the two 8-byte reads stay two ds_read_b64 instead of one ds_read2_b64.  Measured: no gain (forward launches +0..2 %).
// Also measured and not kept: fragments three sub-steps ahead instead of one (rings of 4; 256 VGPRs + 8-32 B scratch): +14 %.
#ifndef NSC_FRAG8_SPLIT
#define NSC_FRAG8_SPLIT 0
#endif
__device__ __forceinline__ bf16x8 ld_frag8(nsc_lds_cu16 p) {
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) const u32x2_* lds_u2;
#if NSC_FRAG8_SPLIT
  unsigned a2 = (unsigned)(unsigned long long)(p + 4);
  asm volatile("" : "+v"(a2));
  const u32x2_ a = *(lds_u2)(p), b = *(lds_u2)(unsigned long long)a2;
#else
  const u32x2_ a = *(lds_u2)(p), b = *(lds_u2)(p + 4);
#endif
  const i32x4_t v = {(int)a[0], (int)a[1], (int)b[0], (int)b[1]};
  return __builtin_bit_cast(bf16x8, v);
}

// ---- geometry shared by the kernel, the launcher and the image builder ----
constexpr int SPL_TT = 64, SPL_LDX = 112;
constexpr int SPL_GCOLS = 80;                               // g plane: 5 column tiles (fresh); rows 72.. are zeros (k9 k-slots 180..191)
constexpr int SPL_GPL = SPL_GCOLS * NARROW;                 // elements per g plane
constexpr int SPL_KS2 = 10, SPL_KS3 = 6;                    // k-steps of 32: k15 (300 -> 320), k9 (180 -> 192)
constexpr int SPL_W2ROWS = 2 * NARROW;                      // rows of the gate image (no row padding: see w2 below)
constexpr int SPL_W2U16 = SPL_KS2 * 3 * 4 * SPL_W2ROWS * 8; // [k-step][plane][q][row 40][8]
constexpr int SPL_W2SLACK = 64;                             // u16: rows 40..47 of the last (k-step, plane, q) block read 128 B past the image
template <int DIL>
struct SplGeom {
  static constexpr int H = 4 + 7 * DIL, WX = SPL_TT + 2 * H, NCT1 = (WX + 15) / 16, HCOLS = NCT1 * 16;
  static constexpr int HPL = HCOLS * NARROW;                // elements per h plane (both parity halves)
  static constexpr int HHALF = HCOLS / 2 * NARROW;          // DIL 2: elements per parity half
};
template <int DIL>
__device__ __forceinline__ int spl_hoff(int j) {            // element offset of h column j in a plane
  return DIL == 1 ? j * NARROW : (j & 1) * SplGeom<DIL>::HHALF + (j >> 1) * NARROW;
}
static size_t spl_fwd_smem(int CR, int dil) {
  const int hpl = dil == 1 ? SplGeom<1>::HPL : SplGeom<2>::HPL;
  return (size_t)CR * SPL_LDX * 4 + 3 * (size_t)hpl * 2 + 3 * (size_t)SPL_GPL * 2 + (size_t)(SPL_W2U16 + SPL_W2SLACK) * 2 +
         (size_t)2 * (CR / 4) * 64 * 4;
}

// -----------------------------------------------------------------------------------------------------
// Forward.  Jobs:
//   phase 1 (fp32, as gated_block_fwd2_body): wave w = row tile w >> 2 of h, column tile w & 3 (+ a second one in a fresh tile)
//   phase 2: 3 row tiles (lin / tanh interleaved: rows 4m + {0,1} = lin of channels 2m, 2m+1 of the tile, 4m + {2,3} = tanh) x 4 | 5
//            column tiles.  Waves 0-3: row tiles 0 and 1 of column tile w (one set of B fragments); waves 4-7: row tile 2 of column
//            tile w - 4, and in a fresh tile waves 4-6 also row tile w - 4 of the fifth column tile: 3 | 3 | 3 | 3 (steady) and
//            4 | 4 | 4 | 3 (fresh) jobs per SIMD.  A fragments from the LDS image w2, B from the h planes.
//   phase 3: a wave owns ONE row tile of 16 output channels (its k9 fragments: 6 k-steps x 3 planes in 72 registers) and walks the
//            column tiles: C = 100: waves 0-6 x 4 column tiles (wave 7: nothing); C <= 50: wave w = row tile w & 3, column tiles
//            2 (w >> 2) + {0, 1}.  Transposed product as in block.hip (a lane's accumulator = 4 consecutive steps of one channel).
// The activations kept for the backward pass leave from the accumulators (64-byte pieces); on this path the matrix pipe does not
// wait for the stores of the partner wave.
// -----------------------------------------------------------------------------------------------------
template <int RT9, int NK1, int DIL, bool PAIRED, bool FIRST = false>
__device__ __forceinline__ void gated_block_fwd3_body(const BlockArgs& a, int ntiles, int tpf, int* flags, int* timeouts) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using G = SplGeom<DIL>;
  constexpr int TT = SPL_TT, H = G::H, WX = G::WX, WGW = TT + 8, LDX = SPL_LDX, CR = 4 * NK1, NCT1 = G::NCT1;
  constexpr int HPL = G::HPL, GPL = SPL_GPL;
  static_assert(NCT1 * 16 <= LDX && 79 + 14 * DIL < NCT1 * 16, "h tile must cover every column the k15 taps read");
  // x tile, fp32 rows.  Physical column 0 of a row is step t0 - H - DLT, a multiple of 4 (DLT in 0..3): the tile lands by LDS-DMA in
  // aligned 16-byte pieces that lie wholly inside or wholly outside the frame (T % 4 == 0); the logical tile starts DLT columns in.
  constexpr int DLT = (4 - (H & 3)) & 3;
  static_assert(WX + DLT <= LDX, "shifted x tile fits its rows");
  float* xs = sm + DLT;                                                  // [CR][LDX] fp32, column j <-> step t0 - H + j
  u16* hp = reinterpret_cast<u16*>(sm + CR * LDX);                      // [3][HPL]   (16-byte aligned: from the physical base)
  u16* gp = hp + 3 * HPL;                                               // [3][GPL]
  u16* w2 = gp + 3 * GPL;                                               // gate image
  float* w1s = reinterpret_cast<float*>(w2 + SPL_W2U16 + SPL_W2SLACK);  // [2][NK1][64] fp32 fragments of W1 (registers are what is scarce)
  static_assert((3 * HPL * 2) % 16 == 0 && (3 * GPL * 2) % 16 == 0, "16-byte aligned LDS regions");
  const int C = a.C, T = a.T;
  const int Cin = a.Cin;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;

  // ---- x tile: global -> LDS by LDS-DMA (buffer form: 16 bytes per lane to m0 + lane * 16, ZEROS for lanes whose offset is out of
  // range - tools/blds_check.hip), no registers, no staging pass.  Float4 i of the [CR][28] tile: row i / 28, piece i % 28; rows
  // past Cin and pieces outside the frame are sent out of range.  The tile of step n + 1 is requested as soon as phase 1 of step n
  // has read the buffer (the residual of phase 3 comes from memory - an L2 hit: this CU fetched the same lines a tile ago).
  constexpr int NX4 = CR * (LDX / 4), NDMA = (NX4 + 511) / 512;
  typedef int rsrc4_t __attribute__((ext_vector_type(4)));
  const unsigned long long xaddr = (unsigned long long)a.x;
  const rsrc4_t sxd = {(int)(unsigned)xaddr, (int)((unsigned)(xaddr >> 32) & 0xFFFFu), (int)(unsigned)((long)a.B * Cin * T * 4), 0x00020000};
  const __amdgpu_buffer_rsrc_t sx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * Cin * T * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (unsigned)((long)a.B * a.C * T * 4), 0x00020000);
  const unsigned lds_x = (unsigned)(unsigned long long)(nsc_lds_cu16)reinterpret_cast<const u16*>(sm);
  auto dma_x = [&](int tile) {
    const int tl = __builtin_amdgcn_readfirstlane(tile < ntiles ? tile : 0);
    const int b = tl / tpf, t0 = (tl - b * tpf) * TT;
#pragma unroll
    for (int n = 0; n < NDMA; ++n) {
      const int i = tid + 512 * n;
      if (i < NX4) {
        const int r = i / (LDX / 4), f = i - r * (LDX / 4);
        const int tm = t0 - H - DLT + 4 * f;
        const int vo = (r < Cin && tm >= 0 && tm < T) ? ((b * Cin + r) * T + tm) * 4 : 0x7ffffff0;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_x + (unsigned)((n * 8 + wave) * 1024));
        if (PAIRED) asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen sc0 sc1 lds" ::"v"(vo), "s"(sxd), "s"(m0v) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(vo), "s"(sxd), "s"(m0v) : "memory", "m0");
      }
    }
  };
  const int first = (int)((long)blockIdx.x * ntiles / gridDim.x), last = (int)((long)(blockIdx.x + 1) * ntiles / gridDim.x);
  NSC_STAMP(32);
  if (!PAIRED) dma_x(first);

  // ---- once per workgroup: the parameter image (nsc_gated_block_simage_index, which = 0) ----
  const int r1 = wave >> 2;
  float b1r[4];
  const int rt3 = RT9 == 7 ? min(wave, 6) : (wave & 3);
  bf16x8 w9a[SPL_KS3][3];
  float b9l;
  float blr[2][2], brr[2][2];
  {
    const f32x4* img4 = reinterpret_cast<const f32x4*>(a.img);
    constexpr int N2_4 = (SPL_W2U16 + SPL_W2SLACK) / 8;                  // 16-byte units of the gate image
    constexpr int NFA = NK1 + 4, NF4A = (NFA + 3) / 4, NF4B = SPL_KS3 * 3 + 1, NF4C = 2, NVB = RT9 == 7 ? 7 : 4;
    f32x4 fa[NF4A], fc[NF4C];
    const f32x4* fbase = img4 + N2_4 + lane;
#pragma unroll
    for (int g = 0; g < NF4A; ++g) fa[g] = fbase[(r1 * NF4A + g) * 64];
    const f32x4* fb = fbase + (2 * NF4A + rt3 * NF4B) * 64;
#pragma unroll
    for (int s = 0; s < SPL_KS3; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p) w9a[s][p] = __builtin_bit_cast(bf16x8, fb[(s * 3 + p) * 64]);
    b9l = fb[SPL_KS3 * 3 * 64][0];
#pragma unroll
    for (int g = 0; g < NF4C; ++g) fc[g] = fbase[(2 * NF4A + NVB * NF4B + wave * NF4C + g) * 64];
    if ((wave & 3) == 0) {                    // W1 fragments of row tile r1 -> LDS (read by the four waves that share the row tile)
#pragma unroll
      for (int u = 0; u < NK1; ++u) w1s[(r1 * NK1 + u) * 64 + lane] = fa[u / 4][u % 4];
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) b1r[reg] = fa[(NK1 + reg) / 4][(NK1 + reg) % 4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        blr[e][u] = fc[0][2 * e + u];
        brr[e][u] = fc[1][2 * e + u];
      }
  }
  if (PAIRED) {
    nsc_pair_wait(flags, timeouts);
    dma_x(first);
  }
  nsc_wait_vmem();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler does not see the DMA of the first x tile)
  // the gate image by LDS-DMA (16 bytes per lane, no registers): the youngest vector-memory operations when the tile loop starts,
  // waited for by hand before the first tile's phase 2 (see gated_block_fwd2_body)
  {
    constexpr int N2_4 = (SPL_W2U16 + SPL_W2SLACK) / 8, NG2 = (N2_4 + 511) / 512;
    const unsigned lds2 = (unsigned)(unsigned long long)w2;
#pragma unroll
    for (int i = 0; i < NG2; ++i) {
      if (tid + 512 * i < N2_4) {
        const f32x4* gptr = reinterpret_cast<const f32x4*>(a.img) + tid + 512 * i;
        const unsigned m0v = __builtin_amdgcn_readfirstlane(lds2 + (unsigned)((i * 8 + wave) * 1024));
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(m0v) : "memory", "m0");
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
      // carried columns (the planes move as 32-bit words): g [64, 72) -> [0, 8); h [72, 2H + 64) -> [8, 2H) (dilation 2: rows
      // [36, H + 32) -> [4, H) of either parity half)
      constexpr int NGW = 8 * NARROW / 2, NHW = (2 * H - 8) * NARROW / 2 / DIL, NHALF = DIL;
      constexpr int NCW = 3 * (NGW + NHALF * NHW), NCI = (NCW + 511) / 512;
      // (all reads first, then all writes: one LDS round trip instead of one per pass of the loop)
      unsigned cv[NCI];
      unsigned* cdst[NCI];
#pragma unroll
      for (int it = 0; it < NCI; ++it) {
        const int e = min(tid + 512 * it, NCW - 1);
        const int p = e / (NGW + NHALF * NHW), r = e - p * (NGW + NHALF * NHW);
        if (r < NGW) {
          cdst[it] = reinterpret_cast<unsigned*>(gp + p * GPL) + r;
          cv[it] = cdst[it][TT * NARROW / 2];
        } else {
          const int rr = r - NGW, hf = rr / NHW, i = rr - hf * NHW;
          cdst[it] = reinterpret_cast<unsigned*>(hp + p * HPL + hf * G::HHALF) + (8 / DIL) * NARROW / 2 + i;
          cv[it] = cdst[it][(TT / DIL) * NARROW / 2];
        }
      }
#pragma unroll
      for (int it = 0; it < NCI; ++it)
        if (tid + 512 * it < NCW) *cdst[it] = cv[it];
    }
    NSC_STAMP(35);
    nsc_lds_barrier();           // the x tile is in LDS (waited for in the previous tile's phase 3 / the prologue), the carried columns moved
    NSC_STAMP(36);

    // ---- phase 1 (exact fp32): h = lrelu(W1 x + b1) -> the three bf16 planes; kept for the backward pass from the accumulators ----
    {
      const int jb0 = fresh ? (wave & 3) * 16 : 2 * H + (wave & 3) * 16;
      const int jb1 = min((wave & 3) + 4, NCT1 - 1) * 16;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float* xc0 = xs + kq * LDX + jb0 + l15;
      const float* xc1 = xs + kq * LDX + jb1 + l15;
      const float* w1l = w1s + r1 * NK1 * 64 + lane;
      if (fresh) {
#pragma unroll
        for (int u = 0; u < NK1; ++u) {
          const float wv = w1l[u * 64];
          acc0 = mfma4(wv, xc0[4 * u * LDX], acc0);
          acc1 = mfma4(wv, xc1[4 * u * LDX], acc1);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NK1; ++u) acc0 = mfma4(w1l[u * 64], xc0[4 * u * LDX], acc0);
      }
      // h is kept for time t in [t_lo, t_hi): a fresh tile from its own first step (the left halo belongs to the previous tile),
      // a steady one from where the previous tile stopped; up to the end of the right halo when the next tile will not recompute it
      const int t_lo = fresh ? t0 : t0 + H, t_hi = min(next_steady ? t0 + H + TT : t0 + TT, T);
      if (r1 == 0 || kq == 0) {                                          // rows 16..19: lane group kq = 0 of the second row tile
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          if (e == 1 && !fresh) break;
          const int j = (e ? jb1 : jb0) + l15;
          const int t = t0 - H + j;
          const bool live = j < WX && t >= 0 && t < T;
          float v[4];
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            float u_ = (e ? acc1[reg] : acc0[reg]) + b1r[reg];
            u_ = u_ > 0.f ? u_ : NSC_LRELU_ALPHA * u_;
            v[reg] = live ? u_ : 0.f;
          }
          unsigned pa[3], pb[3];
          nsc_split2(v[0], v[1], pa);
          nsc_split2(v[2], v[3], pb);
          u16* dst = hp + spl_hoff<DIL>(j) + r1 * 16 + kq * 4;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * HPL) = make_uint2(pa[p], pb[p]);
          if (a.h_out && t >= t_lo && t < t_hi) {
            float* go = a.h_out + ((long)b * NARROW + r1 * 16 + kq * 4) * T + t;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) __builtin_nontemporal_store(v[reg], go + (long)reg * T);
          }
        }
      }
    }
    NSC_STAMP(37);
    if (tile == first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gate image has landed
    nsc_lds_barrier();
    NSC_STAMP(38);
    if (tile + 1 < last) dma_x(tile + 1);     // phase 1 was the last reader of the x tile: the next one lands under phases 2 and 3

    // ---- phase 2: both k15 gate convs on the matrix cores; gate; g -> the three bf16 planes ----
    {
      const int joff = fresh ? 0 : 8;
      f32x4 acc[2];
      acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // image row of this lane in row tile rt: rt * 16 + l15 (rows 40..47 of the third tile: don't-care rows, never stored)
      const u16* wl_ = w2 + (kq * SPL_W2ROWS + l15) * 8;
      int jj0, jj1 = 0, rt0, rt1 = 0;
      bool two;
      // Software-pipelined by hand (hipcc issues a k-step's LDS reads, waits, then its MFMAs - every step exposed a full LDS round
      // trip with the matrix pipe idle: phase 2 ran at 42 % of its MFMA time): the fragments of step s + 1 are requested before
      // the MFMAs of step s issue; scheduling fences keep the order.
      if (wave < 4) {
        rt0 = 0; rt1 = 1; two = true;
        jj0 = jj1 = wave * 16 + l15 + joff;
        // one base per plane, kept opaque: the k-step offsets (64 s bytes) then fit the 8-bit offsets of ds_read2_b64 (with the
        // plane offset folded in, hipcc materialised all 30 addresses in registers)
                nsc_lds_cu16 hb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) hb[p] = nsc_opaque_lds(hp + spl_hoff<DIL>(jj0) + 8 * kq + p * HPL);
        // sub-step u = 2 s + r: row tile r of k-step s.  A fragments one sub-step ahead, B fragments of the next k-step from the
        // odd sub-step: 48 fragment registers instead of the 72 of whole-k-step double buffering
        bf16x8 bb[2][3], aa[2][3];
        auto fetch_a = [&](int u, int slot) {
          const int s_ = u >> 1, r = u & 1;
#pragma unroll
          for (int p = 0; p < 3; ++p) aa[slot][p] = ld_frag16(wl_ + (((NSC_EXP & 1024) ? 0 : s_) * 3 + p) * (4 * SPL_W2ROWS * 8) + r * 16 * 8);
        };
        auto fetch_b = [&](int s_, int slot) {
#pragma unroll
          for (int p = 0; p < 3; ++p) bb[slot][p] = ld_frag8(hb[p] + 32 * ((NSC_EXP & 512) ? 0 : s_));
        };
        fetch_b(0, 0);
        fetch_a(0, 0);
        if (NSC_EXP & 8192) { fetch_b(1, 1); fetch_a(1, 1); }
#pragma unroll
        for (int u = 0; u < 2 * SPL_KS2; ++u) {
          if (!(NSC_EXP & 8192)) {      // (timing experiment 8192: no operand fetches after the first - WRONG RESULTS)
          if (u + 1 < 2 * SPL_KS2) fetch_a(u + 1, (u + 1) & 1);
          if ((u & 1) && (u >> 1) + 1 < SPL_KS2) fetch_b((u >> 1) + 1, ((u >> 1) + 1) & 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[u & 1] = mfma_split6(aa[u & 1], bb[(u >> 1) & 1], acc[u & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        rt0 = 2;
        jj0 = (wave - 4) * 16 + l15 + joff;
        two = fresh && wave < 7;
        rt1 = wave - 4;
        jj1 = 64 + l15;                                                  // the fifth column tile of a fresh tile
        nsc_lds_cu16 hbe[2][3];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int p = 0; p < 3; ++p) hbe[e][p] = nsc_opaque_lds(hp + spl_hoff<DIL>(e ? jj1 : jj0) + 8 * kq + p * HPL);
        const u16* wa1 = wl_ + (two ? rt1 : 0) * 16 * 8;
        bf16x8 bb[2][3], aa[2][3];
        if (two) {
          // sub-step u = 2 s + e: job e of k-step s, fragments one sub-step ahead
          auto fetch = [&](int u, int slot) {
            const int s_ = u >> 1, e = u & 1;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              bb[slot][p] = ld_frag8(hbe[e][p] + 32 * ((NSC_EXP & 512) ? 0 : s_));
              aa[slot][p] = ld_frag16((e ? wa1 : wl_ + 32 * 8) + (((NSC_EXP & 1024) ? 0 : s_) * 3 + p) * (4 * SPL_W2ROWS * 8));
            }
          };
          fetch(0, 0);
          if (NSC_EXP & 8192) fetch(1, 1);
#pragma unroll
          for (int u = 0; u < 2 * SPL_KS2; ++u) {
            if (!(NSC_EXP & 8192) && u + 1 < 2 * SPL_KS2) fetch(u + 1, (u + 1) & 1);
              __builtin_amdgcn_sched_barrier(0);
            acc[u & 1] = mfma_split6(aa[u & 1], bb[u & 1], acc[u & 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          auto fetch = [&](int s_, int slot) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              bb[slot][p] = ld_frag8(hbe[0][p] + 32 * ((NSC_EXP & 512) ? 0 : s_));
              aa[slot][p] = ld_frag16(wl_ + 32 * 8 + (((NSC_EXP & 1024) ? 0 : s_) * 3 + p) * (4 * SPL_W2ROWS * 8));
            }
          };
          fetch(0, 0);
          if (NSC_EXP & 8192) fetch(1, 1);
#pragma unroll
          for (int s_ = 0; s_ < SPL_KS2; ++s_) {
            if (!(NSC_EXP & 8192) && s_ + 1 < SPL_KS2) fetch(s_ + 1, (s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = mfma_split6(aa[s_ & 1], bb[s_ & 1], acc[0]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      NSC_STAMP(44);
      // epilogue: the lane holds (lin c0, lin c0+1, tanh-pre c0, tanh-pre c0+1) of one step
      const int s_lo = fresh ? 4 : 8, s_hi = next_steady ? WGW : 4 + TT;   // columns kept for the backward pass (see save_lg in block.hip)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && !two) continue;
        const int jj = e ? jj1 : jj0, rt = e ? rt1 : rt0;
        const int c0 = rt * 8 + kq * 2;
        const int t = t0 - 4 + jj;
        const bool live = jj < WGW && t >= 0 && t < T;
        if (c0 < NARROW) {
          float lin[2], th[2], gg[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            lin[u] = acc[e][u] + blr[e][u];
            th[u] = nsc_tanh(acc[e][2 + u] + brr[e][u]);
            gg[u] = live ? lin[u] * th[u] : 0.f;
          }
          unsigned pg[3];
          nsc_split2(gg[0], gg[1], pg);
          u16* dst = gp + jj * NARROW + c0;
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(dst + p * GPL) = pg[p];
          if (a.lin_out && live && jj >= s_lo && jj < s_hi) {
            const long gi = ((long)b * NARROW + c0) * T + t;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              __builtin_nontemporal_store(lin[u], a.lin_out + gi + (long)u * T);
              __builtin_nontemporal_store(th[u], a.th_out + gi + (long)u * T);
              __builtin_nontemporal_store(gg[u], a.g_out + gi + (long)u * T);
            }
          }
        }
      }
    }
    NSC_STAMP(39);
    nsc_lds_barrier();
    NSC_STAMP(40);

    // ---- phase 3: y = W9 * g + b9 + x on the matrix cores (transposed product: rows of D = time, columns = output channels) ----
    int lane3 = lane;
    asm volatile("" : "+v"(lane3));
    const int l15p = lane3 & 15, kqp = lane3 >> 4;
    // (round 6: the epilogue's addresses are 32-bit buffer offsets built ONCE per tile - the lane's row, the tile's scalar part - plus
    // constants; it used to form a 64-bit address per store: two quarter-rate multiplies and a 64-bit mad each.  The activation is
    // max(u, slope u), slope = 1 for a flat block: two instructions instead of a compare and two selects.)
    const float slope3 = a.flat ? 1.f : NSC_LRELU_ALPHA;
    auto dense3 = [&](auto nc_c, int ct0) {
      constexpr int NC = decltype(nc_c)::value;
      f32x4 acc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the residual x[o][t .. t + 3] from memory (the LDS tile already belongs to the next step): requested before the MFMA loop
      const int o = rt3 * 16 + l15p;
      const int tt0 = ct0 * 16 + 4 * kqp;                                  // first step of this lane's group in column tile ct0
      const int xrow = ((b * Cin + (NK1 == 1 ? 0 : o)) * T + t0 + tt0) * 4;   // Cin = 1: broadcast residual
      const int orow = ((b * C + o) * T + t0 + tt0) * 4;
      f32x4 xr4[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int vo = (o < C && t0 + tt0 + 16 * c < T) ? xrow + 64 * c : 0x7ffffff0;
        xr4[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sx, vo, 0, PAIRED ? NSC_AUX_COHERENT : 0));
      }
      // (software-pipelined by hand like phase 2: step i = (k-step i / NC, column tile i % NC); opaque bases per plane and pair of
      // column tiles: the offsets of a pair's fragments - up to 640 + 320 + 8 bytes - fit the 8-bit offset fields)
      nsc_lds_cu16 gb[3][(NC + 1) / 2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int c2 = 0; c2 < (NC + 1) / 2; ++c2) gb[p][c2] = nsc_opaque_lds(gp + p * GPL + ((ct0 + 2 * c2) * 16 + l15p) * NARROW + 8 * kqp);
      bf16x8 gf[2][3];
      auto fetch = [&](int i, int slot) {
        const int s = i / NC, c = i - s * NC;
#pragma unroll
        for (int p = 0; p < 3; ++p) gf[slot][p] = ld_frag8(gb[p][(NSC_EXP & 2048) ? 0 : (c >> 1)] + ((NSC_EXP & 2048) ? 0 : ((c & 1) * 16 * NARROW + 32 * s)));
      };
      fetch(0, 0);
#pragma unroll
      for (int i = 0; i < SPL_KS3 * NC; ++i) {
        if (i + 1 < SPL_KS3 * NC) fetch(i + 1, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[i % NC] = mfma_split6(gf[i & 1], w9a[i / NC], acc[i % NC]);   // (A = g: rows of D are time)
        __builtin_amdgcn_sched_barrier(0);
      }
      // Every vector-memory operation this wave has issued so far has completed - the next x tile's DMA (which the compiler does not
      // see) among them: the loop-end barrier then means "the x tile is in LDS".  Everything outstanding here is old (the DMA and the
      // residual went out before ~150 MFMAs), the output stores below are not waited for.
      NSC_STAMP(45);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NSC_STAMP(46);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (o < C && t0 + tt0 + 16 * c < T) {                              // (T % 4 == 0: a lane's four steps are in or out together)
          f32x4 v;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float u_ = acc[c][reg] + b9l + xr4[c][reg];
            v[reg] = fmaxf(u_, slope3 * u_);
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, v), sout, orow + 64 * c, 0, FIRST ? NSC_AUX_COHERENT : 0);
        }
      }
    };
    if (RT9 == 7) {
      if (wave < 7) dense3(std::integral_constant<int, 4>{}, 0);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (wave 7 has no phase-3 job: its share of the DMA)
    } else {
      dense3(std::integral_constant<int, 2>{}, 2 * (wave >> 2));
    }
    NSC_STAMP(41);
    nsc_lds_barrier();   // xs / the planes are rewritten by the next tile
    NSC_STAMP(42);
  }
  NSC_STAMP(43);
}

template <int RT9, int NK1, int DIL>
__global__ __launch_bounds__(512) void gated_block_fwd3_kernel(BlockArgs a, int ntiles, int tpf) {
  gated_block_fwd3_body<RT9, NK1, DIL, false>(a, ntiles, tpf, nullptr, nullptr);
}
template <int RT9, int NK1A, int NK1B>
__global__ __launch_bounds__(512) void gated_block_fwd3_pair_kernel(BlockArgs a0, BlockArgs a1, int ntiles, int tpf, int* flags, int* timeouts) {
  gated_block_fwd3_body<RT9, NK1A, 1, false, true>(a0, ntiles, tpf, nullptr, nullptr);
  nsc_pair_publish(flags);
  gated_block_fwd3_body<RT9, NK1B, 2, true>(a1, ntiles, tpf, flags, timeouts);
}

// =====================================================================================================
// Split parameter images.  An image is an array of 32-bit WORDS built by nsc_gather / nsc_step_begin from the flat parameter
// buffer (misc.hip: gather_word): idx[w] < 0 -> 0; bits 26..29 = 0 -> the fp32 value src[idx]; = m > 0 -> two bf16 PIECES packed,
// low half of src[i], high half of src[i + stride], plane = (m - 1) / 5 (0 hi, 1 lo, 2 lo2), stride = {20, 25, 50, 100, 1}[(m - 1) % 5]
// (the two k of a word are neighbours in the reduction index: the source stride is the kernel's Cout in the forward, 1 in the
// data gradients, whose reduction runs over the output channel of the [K][Cin][Cout] kernels).
//   which = 0 (forward): gate image [10][3][4][40][4 words] (+ 32 words of zeros) | group A [2][nf4a][64][4] fp32 (W1 fragments + b1,
//   as in nsc_gated_block_image_index) | group B per phase-3 row tile: [6][3][64][4 words] k9 fragments + [64][4] (word 0: b9 of the lane's
//   channel) | group C [8][2][64][4] fp32: gate biases of the wave's two phase-2 jobs.
// =====================================================================================================
// which = 2: the weight pieces of the three-launch data gradient (block_bwd_split.hip)
long nsc_bb_simage_words(int C);
void nsc_bb_simage_index(int C, long w9, long wl, long wr, const int mode_bits[3], int* idx);
static bool simg_shape(int which, int C, int Cin, int dil, int* rt9, int* nk) {
  if (which == 2) return (C == 100 || C == 50) && (Cin == C || Cin == 1) && (dil == 1 || dil == 2);
  if (which != 0) return false;                                              // (which = 1, the fused split data gradient of round 5, is gone)
  if (!(dil == 1 || dil == 2) || !(C == 100 || C == 50 || C == 25) || !(Cin == C || Cin == 1)) return false;
  *rt9 = C == 100 ? 7 : 4;
  *nk = C == 100 ? 25 : (C == 50 ? 13 : 7);
  return true;
}
static int simg_mode(int plane, int stride) {
  const int sel = stride == 20 ? 0 : (stride == 25 ? 1 : (stride == 50 ? 2 : (stride == 100 ? 3 : 4)));   // (4: stride 1)
  return (1 + plane * 5 + sel) << 26;
}
extern "C" long nsc_gated_block_simage_words(int which, int C, int Cin, int dil) {
  int rt9, nk;
  if (!simg_shape(which, C, Cin, dil, &rt9, &nk)) return 0;
  if (which == 2) return nsc_bb_simage_words(C);
  const int nk1 = Cin == 1 ? 1 : nk;
  const long n2 = (SPL_W2U16 + SPL_W2SLACK) / 2;
  return n2 + 256L * (2 * ((nk1 + 4 + 3) / 4) + (rt9 == 7 ? 7 : 4) * (SPL_KS3 * 3 + 1) + 8 * 2);
}
extern "C" int nsc_gated_block_simage_index(int which, int C, int Cin, int dil, const long* offs, int* idx) {
  NSC_REQUIRE(offs && idx, NSC_ERR_BAD_ARG, "nsc_gated_block_simage_index: null pointer");
  int rt9, nk;
  NSC_REQUIRE(simg_shape(which, C, Cin, dil, &rt9, &nk), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_simage_index: no split image for C %d, Cin %d, dil %d, which %d", C, Cin, dil, which);
  const long n = nsc_gated_block_simage_words(which, C, Cin, dil);
  for (long i = 0; i < n; ++i) idx[i] = -1;
  auto mn = [](int a_, int b_) { return a_ < b_ ? a_ : b_; };
  const long w1 = offs[0], b1 = offs[1], wl = offs[2], bl = offs[3], wr = offs[4], br = offs[5], w9 = offs[6], b9 = offs[7];
  for (long i = 0; i < 8; ++i)
    NSC_REQUIRE(offs[i] >= 0 && offs[i] < (1L << 26) - (1L << 20), NSC_ERR_UNSUPPORTED, "nsc_gated_block_simage_index: offset %ld does not fit 26 bits", offs[i]);
  if (which == 2) {
    const int modes[3] = {simg_mode(0, 1), simg_mode(1, 1), simg_mode(2, 1)};
    nsc_bb_simage_index(C, w9, wl, wr, modes, idx);
    return NSC_OK;
  }
  const int nk1 = Cin == 1 ? 1 : nk;
  // gate image: row r of the 40 -> tile r / 16, ii = r % 16: channel (r / 16) * 8 + (ii >> 2) * 2 + (ii & 1), branch (ii & 2) ? tanh : lin
  for (int s = 0; s < SPL_KS2; ++s)
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 4; ++q)
        for (int r = 0; r < SPL_W2ROWS; ++r)
          for (int jw = 0; jw < 4; ++jw) {
            const int k = 32 * s + 8 * q + 2 * jw, ii = r & 15;
            const int c = (r >> 4) * 8 + (ii >> 2) * 2 + (ii & 1);
            const long wi = ((((long)(s * 3 + p) * 4 + q) * SPL_W2ROWS + r) * 4 + jw);
            if (k < K15 * NARROW) idx[wi] = (int)(((ii & 2) ? wr : wl) + (long)k * NARROW + c) | simg_mode(p, NARROW);
          }
  const int nfa = nk1 + 4, nf4a = (nfa + 3) / 4, nf4b = SPL_KS3 * 3 + 1, nf4c = 2, nvb = rt9 == 7 ? 7 : 4;
  const long baseA = (SPL_W2U16 + SPL_W2SLACK) / 2, baseB = baseA + 2L * nf4a * 256, baseC = baseB + (long)nvb * nf4b * 256;
  for (int lane = 0; lane < 64; ++lane) {
    const int l15 = lane & 15, kq = lane >> 4;
    for (int r1 = 0; r1 < 2; ++r1)
      for (int f = 0; f < nfa; ++f) {
        const long src = f < nk1 ? w1 + (long)mn(4 * f + kq, Cin - 1) * NARROW + mn(r1 * 16 + l15, NARROW - 1)
                                 : b1 + mn(r1 * 16 + kq * 4 + (f - nk1), NARROW - 1);
        idx[baseA + ((long)(r1 * nf4a + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
      }
    for (int rt3 = 0; rt3 < nvb; ++rt3) {
      const int o = mn(rt3 * 16 + l15, C - 1);                 // (columns past C: a finite stand-in, never stored)
      for (int s = 0; s < SPL_KS3; ++s)
        for (int p = 0; p < 3; ++p)
          for (int jw = 0; jw < 4; ++jw) {
            const int k = 32 * s + 8 * kq + 2 * jw;
            if (k < K9 * NARROW)
              idx[baseB + ((long)(rt3 * nf4b + s * 3 + p) * 64 + lane) * 4 + jw] = (int)(w9 + (long)k * C + o) | simg_mode(p, C);
          }
      idx[baseB + ((long)(rt3 * nf4b + SPL_KS3 * 3) * 64 + lane) * 4] = (int)(b9 + o);
    }
    for (int wave = 0; wave < 8; ++wave) {                     // group C: gate biases of the wave's two phase-2 jobs
      const int jrt[2] = {wave < 4 ? 0 : 2, wave < 4 ? 1 : mn(wave - 4, 2)};
      for (int f = 0; f < 8; ++f) {
        const int t = f & 3;
        const long src = (f < 4 ? bl : br) + mn(jrt[t >> 1] * 8 + kq * 2 + (t & 1), NARROW - 1);
        idx[baseC + ((long)(wave * nf4c + f / 4) * 64 + lane) * 4 + (f & 3)] = (int)src;
      }
    }
  }
  return NSC_OK;
}

static int spl_cu_count() {
  int dv = 0, n = 0;
  if (hipGetDevice(&dv) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dv) != hipSuccess) return 0;
  return n;
}

template <int RT9, int NK1, int DIL>
static int launch_block_fwd3(const BlockArgs& a, hipStream_t st) {
  const size_t smem = spl_fwd_smem(4 * NK1, DIL);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "gated_block_fwd3: %zu B LDS", smem);
  auto kern = gated_block_fwd3_kernel<RT9, NK1, DIL>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd3: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a.T, 64);
  const int ntiles = a.B * tpf;
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, 256)), dim3(512), smem, st, a, ntiles, tpf);
  NSC_CHECK_LAUNCH("gated_block_fwd3");
  return NSC_OK;
}
template <int RT9, int NK1A, int NK1B>
static int launch_block_fwd3_pair(const BlockArgs& a0, const BlockArgs& a1, int* flags, int* timeouts, hipStream_t st) {
  const size_t smem = spl_fwd_smem(4 * (NK1A > NK1B ? NK1A : NK1B), 2);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "gated_block_fwd3_pair: %zu B LDS", smem);
  auto kern = gated_block_fwd3_pair_kernel<RT9, NK1A, NK1B>;
  const hipError_t e = NSC_SMEM_ATTR(kern, (int)smem);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_fwd3_pair: smem attr: %s", hipGetErrorString(e));
  const int tpf = nsc_cdiv(a0.T, 64);
  const int ntiles = a0.B * tpf;
  const int grid = std::min(ntiles, 256);
  static std::atomic<int> ncu{0};
  int cu = ncu.load(std::memory_order_relaxed);
  if (cu == 0) { cu = spl_cu_count(); ncu.store(cu, std::memory_order_relaxed); }
  NSC_REQUIRE(grid <= cu, NSC_ERR_UNSUPPORTED, "gated_block_fwd3_pair: %d workgroups > %d CUs", grid, cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, st, a0, a1, ntiles, tpf, flags, timeouts);
  NSC_CHECK_LAUNCH("gated_block_fwd3_pair");
  return NSC_OK;
}

// nsc_gated_block_fwd_img on a SPLIT image (nsc_gated_block_simage_index, which = 0): same arguments, same results to fp32 rounding
extern "C" int nsc_gated_block_fwd_simg(const float* img, const float* x, float* out, float* h_out, float* lin_out, float* th_out,
                                        float* g_out, int B, int C, int Cin, int T, int dil, int flat, void* stream) {
  NSC_REQUIRE(img && x && out, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_simg: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_simg: bad sizes");
  NSC_REQUIRE(nsc_gated_block_simage_words(0, C, Cin, dil) > 0, NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_fwd_simg: no split kernel for C %d, Cin %d, dil %d", C, Cin, dil);
  NSC_REQUIRE(((uintptr_t)img & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_fwd_simg: image must be 16-byte aligned");
  NSC_REQUIRE((T & 3) == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0 && (long)B * C * T * 4 < (1L << 31), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_fwd_simg: needs T %% 4 == 0, 16-byte aligned x / out and tensors below 2 GB (T %d, B %d): use nsc_gated_block_fwd_img", T, B);
  NSC_REQUIRE(!(lin_out || th_out || g_out) || (lin_out && th_out && g_out), NSC_ERR_BAD_ARG,
              "nsc_gated_block_fwd_simg: lin/th/g outputs must be given together");
  BlockArgs a{B, C, T, dil, flat, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, h_out, lin_out,
              th_out, g_out, Cin, img};
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 1) {
    if (C == 100) return dil == 1 ? launch_block_fwd3<7, 1, 1>(a, st) : launch_block_fwd3<7, 1, 2>(a, st);
    return dil == 1 ? launch_block_fwd3<4, 1, 1>(a, st) : launch_block_fwd3<4, 1, 2>(a, st);
  }
  if (C == 100) return dil == 1 ? launch_block_fwd3<7, 25, 1>(a, st) : launch_block_fwd3<7, 25, 2>(a, st);
  if (C == 25) return dil == 1 ? launch_block_fwd3<4, 7, 1>(a, st) : launch_block_fwd3<4, 7, 2>(a, st);
  return dil == 1 ? launch_block_fwd3<4, 13, 1>(a, st) : launch_block_fwd3<4, 13, 2>(a, st);
}

// nsc_gated_block_pair_fwd_img on split images
extern "C" int nsc_gated_block_pair_fwd_simg(const float* img0, const float* img1, const float* x, float* out0, float* h0, float* lin0,
                                             float* th0, float* g0, float* out1, float* h1, float* lin1, float* th1, float* g1, int B,
                                             int C, int Cin0, int T, int flat1, int* flags, int* timeouts, void* stream) {
  NSC_REQUIRE(img0 && img1 && x && out0 && out1 && flags && timeouts, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_simg: null pointer");
  NSC_REQUIRE(B > 0 && T > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_simg: bad sizes");
  NSC_REQUIRE(C == 100 || C == 50 || C == 25, NSC_ERR_UNSUPPORTED, "nsc_gated_block_pair_fwd_simg: C %d", C);
  NSC_REQUIRE((T & 3) == 0 && (long)B * C * T * 4 < (1L << 31), NSC_ERR_UNSUPPORTED,
              "nsc_gated_block_pair_fwd_simg: needs T %% 4 == 0 and a tensor below 2 GB (T %d, B %d): launch the blocks one by one", T, B);
  NSC_REQUIRE((((uintptr_t)img0 | (uintptr_t)img1) & 15) == 0, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_simg: images must be 16-byte aligned");
  NSC_REQUIRE((((uintptr_t)x | (uintptr_t)out0 | (uintptr_t)out1) & 15) == 0, NSC_ERR_UNSUPPORTED, "nsc_gated_block_pair_fwd_simg: x / out must be 16-byte aligned");
  NSC_REQUIRE((!(lin0 || th0 || g0) || (lin0 && th0 && g0)) && (!(lin1 || th1 || g1) || (lin1 && th1 && g1)), NSC_ERR_BAD_ARG,
              "nsc_gated_block_pair_fwd_simg: lin/th/g outputs must be given together");
  NSC_REQUIRE(Cin0 == C || Cin0 == 1, NSC_ERR_BAD_ARG, "nsc_gated_block_pair_fwd_simg: Cin0 must be C or 1 (got %d)", Cin0);
  BlockArgs a0{B, C, T, 1, 0, x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out0, h0, lin0, th0, g0, Cin0, img0};
  BlockArgs a1{B, C, T, 2, flat1, out0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out1, h1, lin1, th1, g1, C, img1};
  hipStream_t st = (hipStream_t)stream;
  if (Cin0 == 1) {
    if (C == 100) return launch_block_fwd3_pair<7, 1, 25>(a0, a1, flags, timeouts, st);
    if (C == 25) return launch_block_fwd3_pair<4, 1, 7>(a0, a1, flags, timeouts, st);
    return launch_block_fwd3_pair<4, 1, 13>(a0, a1, flags, timeouts, st);
  }
  if (C == 100) return launch_block_fwd3_pair<7, 25, 25>(a0, a1, flags, timeouts, st);
  if (C == 25) return launch_block_fwd3_pair<4, 7, 7>(a0, a1, flags, timeouts, st);
  return launch_block_fwd3_pair<4, 13, 13>(a0, a1, flags, timeouts, st);
}

// =====================================================================================================
// Weight gradients of the gated blocks on the bf16 matrix cores (split operands): the batched persistent kernel of block.hip
// (gated_block_wgrad_batch_kernel: same job table, same slabs, same flush layout) with k = TIME in steps of 32.
//   dW9[(tap,ci)][o]  = sum_t g[ci][t + tap - 4] dy[o][t]          dWl | dWr[(tap,ci)][c'] = sum_t h[ci][t + (tap - 7) d] da[c'][t]
//   dW1[ci][o]        = sum_t x[ci][t] dz1[o][t]                    bias gradients: a fragment of ones against dy / da / dz1
// Operands whose row is a plain channel (dy, da, dz1, x) are staged as [channel][time] bf16 planes (row pitch 72: 16-byte rows, the 16
// rows a quarter wave reads are 36 banks apart) and read as 16-byte fragments.  The SHIFTED operands (g for the nine taps, h for the
// fifteen) are staged as [time][20] planes with a row pitch of exactly 20, so that row m = tap * 20 + ci of the virtual im2col matrix
// at time t is element t * 20 + m of the plane (header of this file); the A fragment wants 8 consecutive TIMES of one row - the
// transpose of what is contiguous - and comes from two ds_read_b64_tr_b16 (a 4 x 16 block delivered column-major: lane 4 q + p of a
// 16-lane group supplies the address of row q, columns 4 p .. 4 p + 3; lane i receives column i).  Dilation 2: parity halves as in the
// forward.  Rows m >= 180 | 300 of the last row tile and columns past the last channel read whatever follows (rows and columns of a
// product are independent) and are never stored.
// Needs T % 4 == 0 and 16-byte aligned tensors (the staging moves 16 bytes per lane); other shapes take the exact kernel.
// =====================================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ld_frag_tr(const u16* p0, const u16* p1) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  return (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
}
constexpr int WSP_LDT = 72;                                   // row pitch (elements) of the [channel][time] planes
constexpr int WSP_GROWS = 74, WSP_HROWS = 96, WSP_HHALF = 48 * NARROW;
static size_t wsp_smem(int part, int C, int Cx) {
  if (part == 1) return (size_t)2 * (3 * (size_t)C * WSP_LDT + 3 * (size_t)WSP_GROWS * NARROW);
  return (size_t)2 * (3 * (size_t)(Cx + 3 * NARROW) * WSP_LDT + 3 * (size_t)WSP_HROWS * NARROW);
}

// four consecutive steps of one channel -> the three [channel][time] planes (8 bytes per plane)
__device__ __forceinline__ void wsp_put_ct(u16* plane0, int plstride, int off, const f32x4& v) {
  unsigned pa[3], pb[3];
  nsc_split2(v[0], v[1], pa);
  nsc_split2(v[2], v[3], pb);
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(plane0 + p * plstride + off) = make_uint2(pa[p], pb[p]);
}
// four consecutive steps of one channel -> the three [time][20] planes (element offsets o0..o3 of the four steps)
__device__ __forceinline__ void wsp_put_tc(u16* plane0, int plstride, int o0, int o1, int o2, int o3, const f32x4& v) {
  unsigned pa[3], pb[3];
  nsc_split2(v[0], v[1], pa);
  nsc_split2(v[2], v[3], pb);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    u16* pl = plane0 + p * plstride;
    pl[o0] = (u16)(pa[p] & 0xffffu);
    pl[o1] = (u16)(pa[p] >> 16);
    pl[o2] = (u16)(pb[p] & 0xffffu);
    pl[o3] = (u16)(pb[p] >> 16);
  }
}

template <int RT9, int PART>
__device__ __forceinline__ void block_wgrad_split_body(const BlockWgradArgs& a, int wg, int nwg, int slab_id) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  u16* sm = reinterpret_cast<u16*>(smf);
  constexpr int TT = 64, LDT = WSP_LDT;
  const int C = a.C, Cx = a.Cin, T = a.T, d = a.dil;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3
  const int l15 = lane & 15, kq = lane >> 4;
  const int tq = l15 >> 2, tp = l15 & 3;                              // row / column group of this lane in a transposed 4 x 16 block
  // ---- LDS ----
  const int PLY = C * LDT, PLG = WSP_GROWS * NARROW;                  // part 1: dy planes, g planes
  const int PLX = Cx * LDT, PLA = 2 * NARROW * LDT, PLZ = NARROW * LDT, PLH = WSP_HROWS * NARROW;   // part 2
  u16* dyp = sm;
  u16* gpl = sm + 3 * PLY;
  u16* xp = sm;
  u16* dap = xp + 3 * PLX;
  u16* dzp = dap + 3 * PLA;
  u16* hpl = dzp + 3 * PLZ;
  // ---- accumulators: row tiles {w, w + 4, ...} of every product ----
  constexpr int R9 = PART == 1 ? 3 : 1, NC9 = PART == 1 ? RT9 : 1, RLR = PART == 2 ? 5 : 1, R1 = PART == 2 ? (RT9 + 3) / 4 : 1;
  f32x4 g9[R9][NC9], glr[RLR][3], g1[R1][2];
  // bias gradients (sums over time of dy | da | dz1): every thread adds up the float4 pieces it stages (rows crow + 16 q), the 16
  // threads of a row meet once at the end - exact fp32, and no wave-dependent branch inside the MFMA loop
  float bsum[RT9], bsa[3], bsz[2];
#pragma unroll
  for (int q = 0; q < RT9; ++q) bsum[q] = 0.f;
  bsa[0] = bsa[1] = bsa[2] = bsz[0] = bsz[1] = 0.f;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R9; ++r)
#pragma unroll
    for (int c = 0; c < NC9; ++c) g9[r][c] = z4;
#pragma unroll
  for (int r = 0; r < RLR; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) glr[r][c] = z4;
#pragma unroll
  for (int r = 0; r < R1; ++r) g1[r][0] = g1[r][1] = z4;

  // ---- staging: the next tile's windows travel as 16-byte loads into registers during this tile's MFMA loop ----
  const int OOB = 0x7ffffff0;
  const unsigned nbC = (unsigned)((long)a.B * C * T * 4), nbN = (unsigned)((long)a.B * NARROW * T * 4);
  const __amdgpu_buffer_rsrc_t sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((long)a.B * Cx * T * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t sy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, nbC, 0x00020000);
  const __amdgpu_buffer_rsrc_t sa = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.da), 0, 2 * nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz1), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, nbN, 0x00020000);
  const __amdgpu_buffer_rsrc_t sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.h), 0, nbN, 0x00020000);
  auto bl4 = [](const __amdgpu_buffer_rsrc_t& r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  };
  // shifted windows: g starts 4 steps before the tile (18 float4 per channel); h starts HS = 8 d steps before it (a multiple of 4:
  // the first tap reads 7 d steps back), HW4 float4 per channel
  const int HS = 8 * d, HW4 = 16 + 4 * d + (d == 1 ? 0 : 0);          // d = 1: 80 columns, d = 2: 96
  const int WN4 = PART == 1 ? 18 : HW4, WS = PART == 1 ? 4 : HS;
  f32x4 rc[RT9], ra[3], rz[2], rw[2];                                 // rc: dy (part 1) | x (part 2); rw: the g | h window
  const int crow = tid >> 4, cf4 = tid & 15;                          // [channel][time] windows: row crow + 16 q, float4 cf4
  int wrow[2], wf4[2];                                                // window items tid, tid + 256: channel, float4 index
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + 256 * q;
    wrow[q] = i / WN4;
    wf4[q] = i - wrow[q] * WN4;
  }
  auto load_tile = [&](int tile) {
    const int b = __builtin_amdgcn_readfirstlane(tile / a.tiles_per_frame);
    const int t0 = __builtin_amdgcn_readfirstlane((tile - b * a.tiles_per_frame) * TT);
    const int t = t0 + 4 * cf4;
    const int vt = t < T ? (crow * T + t) * 4 : OOB;
    const int rows = PART == 1 ? C : Cx;
#pragma unroll
    for (int q = 0; q < RT9; ++q) {
      const int vo = (crow + 16 * q >= rows) ? OOB : vt;             // (only the last q can pass the last row)
      rc[q] = PART == 1 ? bl4(sy, vo, (b * C + 16 * q) * T * 4) : bl4(sx, vo, (b * Cx + 16 * q) * T * 4);
    }
    if (PART == 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q) ra[q] = bl4(sa, (crow + 16 * q >= 2 * NARROW) ? OOB : vt, (b * 2 * NARROW + 16 * q) * T * 4);
#pragma unroll
      for (int q = 0; q < 2; ++q) rz[q] = bl4(sz, (crow + 16 * q >= NARROW) ? OOB : vt, (b * NARROW + 16 * q) * T * 4);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int tw = t0 - WS + 4 * wf4[q];
      const int vo = (wrow[q] < NARROW && tw >= 0 && tw < T) ? (wrow[q] * T + tw) * 4 : OOB;
      rw[q] = PART == 1 ? bl4(sg, vo, b * NARROW * T * 4) : bl4(sh, vo, b * NARROW * T * 4);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int q = 0; q < RT9; ++q) {
      const int r = crow + 16 * q;
      if (r < (PART == 1 ? C : Cx)) wsp_put_ct(PART == 1 ? dyp : xp, PART == 1 ? PLY : PLX, r * LDT + 4 * cf4, rc[q]);
      if (PART == 1) bsum[q] += (rc[q][0] + rc[q][1]) + (rc[q][2] + rc[q][3]);      // (rows past C were not fetched: zeros)
    }
    if (PART == 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (crow + 16 * q < 2 * NARROW) wsp_put_ct(dap, PLA, (crow + 16 * q) * LDT + 4 * cf4, ra[q]);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (crow + 16 * q < NARROW) wsp_put_ct(dzp, PLZ, (crow + 16 * q) * LDT + 4 * cf4, rz[q]);
#pragma unroll
      for (int q = 0; q < 3; ++q) bsa[q] += (ra[q][0] + ra[q][1]) + (ra[q][2] + ra[q][3]);
#pragma unroll
      for (int q = 0; q < 2; ++q) bsz[q] += (rz[q][0] + rz[q][1]) + (rz[q][2] + rz[q][3]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (wrow[q] < NARROW) {
        const int j = 4 * wf4[q], c = wrow[q];
        if (PART == 1) {
          wsp_put_tc(gpl, PLG, j * NARROW + c, (j + 1) * NARROW + c, (j + 2) * NARROW + c, (j + 3) * NARROW + c, rw[q]);
        } else if (d == 1) {
          wsp_put_tc(hpl, PLH, j * NARROW + c, (j + 1) * NARROW + c, (j + 2) * NARROW + c, (j + 3) * NARROW + c, rw[q]);
        } else {                                                      // parity halves: column j -> (j & 1) * HHALF + (j >> 1) * 20
          const int e0 = (j >> 1) * NARROW + c;
          wsp_put_tc(hpl, PLH, e0, WSP_HHALF + e0, e0 + NARROW, WSP_HHALF + e0 + NARROW, rw[q]);
        }
      }
    }
  };

  // ---- fragment addresses ----
  // [channel][time] planes (B operands, and x as the A operand of dW1): row tile * 16 + l15, k = 32 s + 8 kq + j
  const int ctb = l15 * LDT + 8 * kq;
  // transposed reads: block row tq = time 32 s + 8 kq + tq (+ 4 for the second read), columns m0 + 4 tp ..
  //   g (row 0 <-> step t0 - 4; A[(tap,ci)][tl] = g[ci][t0 + tl + tap - 4] = element (tl + tap) * 20 + ci = tl * 20 + m)
  const int gtr = (8 * kq + tq) * NARROW + 4 * tp + 16 * wave;
  //   h, dilation 1 (row 0 <-> t0 - 8: A = h[ci][t0 + tl + tap - 7] = element (tl + tap + 1) * 20 + ci)
  //   h, dilation 2 (column j <-> t0 - 16 + j: A = column tl + 2 tap + 2 -> parity tl & 1, half row (tl >> 1) + tap + 1)
  const int htr = (d == 1 ? (8 * kq + tq + 1) * NARROW : (tq & 1) * WSP_HHALF + (4 * kq + (tq >> 1) + 1) * NARROW) + 4 * tp + 16 * wave;
  const int hks = d == 1 ? 32 * NARROW : 16 * NARROW, hhf = d == 1 ? 4 * NARROW : 2 * NARROW;   // per k-step / second read

  if (wg < a.ntiles) load_tile(wg);
  for (int tile = wg; tile < a.ntiles; tile += nwg) {
    __syncthreads();                       // everyone is done reading the previous tile
    store_tile();
    __syncthreads();
    if (tile + nwg < a.ntiles) load_tile(tile + nwg);   // in flight during the MFMA loop below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if constexpr (PART == 1) {
        bf16x8 af[R9][3];
#pragma unroll
        for (int r = 0; r < R9; ++r)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u16* q0 = gpl + p * PLG + gtr + s * (32 * NARROW) + r * 64;
            af[r][p] = ld_frag_tr(q0, q0 + 4 * NARROW);
          }
#pragma unroll
        for (int c = 0; c < RT9; ++c) {
          bf16x8 bf[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) bf[p] = ld_frag16(dyp + p * PLY + c * 16 * LDT + ctb + 32 * s);
#pragma unroll
          for (int r = 0; r < R9; ++r) g9[r][c] = mfma_split6(af[r], bf, g9[r][c]);
        }
      } else {
        bf16x8 bd[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) bd[c][p] = ld_frag16(dap + p * PLA + c * 16 * LDT + ctb + 32 * s);
#pragma unroll
        for (int r = 0; r < RLR; ++r) {
          // (19 row tiles of (tap, ci): the twentieth - wave 3, r = 4 - reads rows past the last tap: a product nobody stores, and
          // a loop without a wave-dependent branch)
          bf16x8 af[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const u16* q0 = hpl + p * PLH + htr + s * hks + r * 64;
            af[p] = ld_frag_tr(q0, q0 + hhf);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) glr[r][c] = mfma_split6(af, bd[c], glr[r][c]);
        }
        bf16x8 bz[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) bz[c][p] = ld_frag16(dzp + p * PLZ + c * 16 * LDT + ctb + 32 * s);
#pragma unroll
        for (int r = 0; r < R1; ++r) {                                // (row tiles past Cx: products nobody stores)
          bf16x8 af[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) af[p] = ld_frag16(xp + p * PLX + (wave + 4 * r) * 16 * LDT + ctb + 32 * s);
#pragma unroll
          for (int c = 0; c < 2; ++c) g1[r][c] = mfma_split6(af, bz[c], g1[r][c]);
        }
      }
    }
  }

  // ---- flush: plain stores into this workgroup's private slab (summed afterwards by slab_reduce_batch_kernel) ----
  const long so = (long)slab_id * a.slab_stride;
  if constexpr (PART == 1) {
#pragma unroll
    for (int r = 0; r < R9; ++r)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int kk = (wave + 4 * r) * 16 + kq * 4 + reg;
        if (kk >= K9 * NARROW) continue;
#pragma unroll
        for (int cc = 0; cc < RT9; ++cc) {
          const int o = cc * 16 + l15;
          if (o < C) a.dw9[so + (long)kk * C + o] = g9[r][cc][reg];
        }
      }
#pragma unroll
    for (int q = 0; q < RT9; ++q) {                                   // db9: the 16 threads of a row (lanes 16 g .. 16 g + 15)
      float v = bsum[q];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (cf4 == 0 && crow + 16 * q < C) a.db9[so + crow + 16 * q] = v;
    }
  } else {
#pragma unroll
    for (int r = 0; r < RLR; ++r)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int kk = (wave + 4 * r) * 16 + kq * 4 + reg;
        if (kk >= K15 * NARROW) continue;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
          const int c = ct * 16 + l15;
          if (c < NARROW) a.dwl[so + kk * NARROW + c] = glr[r][ct][reg];
          else if (c < 2 * NARROW) a.dwr[so + kk * NARROW + c - NARROW] = glr[r][ct][reg];
        }
      }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float v = bsa[q];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      const int c = crow + 16 * q;
      if (cf4 == 0 && c < NARROW) a.dbl[so + c] = v;
      else if (cf4 == 0 && c < 2 * NARROW) a.dbr[so + c - NARROW] = v;
    }
#pragma unroll
    for (int r = 0; r < R1; ++r)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ci = (wave + 4 * r) * 16 + kq * 4 + reg;
        if (ci >= Cx) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int o = c * 16 + l15;
          if (o < NARROW) a.dw1[so + ci * NARROW + o] = g1[r][c][reg];
        }
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v = bsz[q];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (cf4 == 0 && crow + 16 * q < NARROW) a.db1[so + crow + 16 * q] = v;
    }
  }
}

template <int RT9>
__global__ __launch_bounds__(256, 2) void gated_block_wgrad_split_batch_kernel(BlockWgradBatch t) {
  const int part = 1 + (blockIdx.x & 1), w = blockIdx.x >> 1;
  int j = 0;
  while (j + 1 < t.njobs && w >= t.wg0[j + 1]) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const BlockWgradArgs& a = t.a[j];
  const int wl = w - t.wg0[j], nw = t.wg0[j + 1] - t.wg0[j];
  if (part == 1) block_wgrad_split_body<RT9, 1>(a, wl, nw, w);
  else block_wgrad_split_body<RT9, 2>(a, wl, nw, w);
}

bool nsc_block_wgrad_split_ok(const BlockWgradBatch& t) {
  for (int q = 0; q < t.njobs; ++q) {
    const BlockWgradArgs& a = t.a[q];
    if ((a.T & 3) || !(a.dil == 1 || a.dil == 2) || !(a.Cin == a.C || a.Cin == 1)) return false;
    const uintptr_t al = (uintptr_t)a.x | (uintptr_t)a.h | (uintptr_t)a.g | (uintptr_t)a.dy | (uintptr_t)a.da | (uintptr_t)a.dz1;
    if (al & 15) return false;
    if ((long)a.B * a.C * a.T * 4 >= (1L << 31)) return false;       // 32-bit buffer offsets
    if (wsp_smem(1, a.C, a.Cin) > 80 * 1024 || wsp_smem(2, a.C, a.Cin) > 80 * 1024) return false;
  }
  return t.njobs > 0;
}
int nsc_launch_block_wgrad_split(const BlockWgradBatch& t, int rt9, int nwg, hipStream_t st) {
  size_t smem = 0;
  for (int q = 0; q < t.njobs; ++q) smem = std::max(smem, std::max(wsp_smem(1, t.a[q].C, t.a[q].Cin), wsp_smem(2, t.a[q].C, t.a[q].Cin)));
  NSC_REQUIRE(smem <= 80 * 1024, NSC_ERR_UNSUPPORTED, "gated_block_wgrad_split: %zu B LDS", smem);
  if (rt9 == 7) {
    auto kern = gated_block_wgrad_split_batch_kernel<7>;
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_wgrad_split: smem attr: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(2 * nwg), dim3(256), smem, st, t);
  } else {
    auto kern = gated_block_wgrad_split_batch_kernel<4>;
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "gated_block_wgrad_split: smem attr: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(2 * nwg), dim3(256), smem, st, t);
  }
  NSC_CHECK_LAUNCH("gated_block_wgrad_split");
  return NSC_OK;
}
