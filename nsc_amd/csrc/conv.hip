// conv.hip - dilated/strided Conv1d (+bias +residual +activation) as implicit GEMM on the CDNA4 matrix
// cores, exact fp32 (v_mfma_f32_16x16x4_f32: bitwise an fmaf chain), layout [B, C, T] (time contiguous).
//
// Replaces tf.compat.v1.layers.conv1d behind nn_core_operator.py:6-14 (reference), forward, data-gradient
// (same kernel on flipped/transposed weights) and weight-gradient.
//
// MFMA mapping (guide: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=4*(l>>4)+reg):
//   forward : rows i = output channel, cols j = time, k = (tap, ci) ; A = weights (global/L2), B = x tile (LDS)
//   wgrad   : rows i = (tap, ci),      cols j = output channel, k = time ; A = x tile (LDS), B = dz tile (LDS)
// so D's lane index is always the memory-contiguous axis of the output (time for y, Cout for dW).
#include "nsc_common.h"
// whole-row 16-byte stores of the conv epilogues are nontemporal (conv class 0.416 -> 0.400 ms per step, profiles/r04h_store_flavours.txt;
// nontemporal 4-byte slab stores of the weight-gradient flush were 6 % slower)
#define NSC_CONV_ST4(p_, v_) __builtin_nontemporal_store(v_, p_)
#include "quant_common.h"
#include <type_traits>
#include <cstdlib>
#include <cstdarg>
#include <algorithm>

// ------------------------------------------------------------------------------------------------
// error plumbing (shared by all translation units)
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void nsc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* nsc_last_error(void) { return g_err; }
extern "C" int nsc_version(void) { return 106; }   // 100 + the round of the last ABI change (6: + nsc_gated_block_dgrad_simg2, nsc_step_begin_chunks, - nsc_gated_block_dgrad_simg; gather index bits 26..29 reserved for word modes since 105)

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup = 4 waves; each wave owns NC column tiles (16 time steps each) and all RT row tiles
// (16 output channels each) of one frame.  x tile (with halo) is staged once in LDS; weight fragments
// stream from global (identical for every wave and workgroup -> L1/L2 resident), prefetched one
// k-step ahead so the MFMA chain of step s covers the latency of step s+1's loads.
// KS = 2: eight waves; waves 4-7 own the same output tiles as waves 0-3 but the ODD taps of the k-loop (intra-workgroup
// split-K): twice the waves in flight per tile and half the dependent MFMA chain per wave; partial sums meet in LDS.
#ifndef NSC_CONV_U
#define NSC_CONV_U 8    // x-tile rows in flight per wave while staging (16 measured no faster: the weight fetch, not staging, bounds these)
#endif
// NPRE > 0: the k-loop has exactly NPRE k-steps (known shapes: 1x1 100->100, k55 1->C) and ALL of a wave's weight
// fragments are fetched into registers up front, in flight together with the x-tile staging; the streamed form pays
// an L2 round trip per group of k-steps, which at one wave per SIMD is most of the kernel for these short reductions.
// Row-wise copy-out of an output tile os [rows][LDO] (row = output channel rt0*16 + r, column = time t0 + j) that the
// kernels below have transposed into LDS: a wave handles whole 256-B rows, so y, the residual and aux move as full cache
// lines and the per-channel bias is a scalar.  NWV waves; TTc columns.
template <int TTc, int NWV>
__device__ __forceinline__ void conv_store_rows(const nsc_conv_desc& d, const float* __restrict__ os, int LDO, int rows_here,
                                                int rt0, int b, int t0, int wave8, int lane, const float* __restrict__ bias,
                                                const float* __restrict__ res, const float* __restrict__ aux,
                                                float* __restrict__ y) {
  const int Cout = d.Cout;
  const int nrow = min(rows_here, Cout - rt0 * 16);
  if (d.out_mode == 1) {
    // sub-pixel shuffle: output row (ch >> 1) interleaves LDS rows 2c, 2c+1 in time, so a wave still writes whole lines;
    // residual / aux (if any) are laid out like the OUTPUT [B, Cout/2, 2 Tout]
    if ((d.Tout & 1) == 0 && (LDO & 1) == 0 && ((uintptr_t)y & 15) == 0 && (!d.res_mode || ((uintptr_t)res & 15) == 0) &&
        (!d.mul_mode || ((uintptr_t)aux & 15) == 0)) {
      // 16 bytes per lane: four consecutive output steps = two steps of LDS row 2c interleaved with two of row 2c + 1
      constexpr int L4 = 2 * TTc / 4, RPI = 64 / L4 > 0 ? 64 / L4 : 1;       // float4 per output row (2 TTc steps)
      static_assert(L4 <= 64, "an output row's float4s fit a wave");
      const int sub = lane / L4, q4 = lane - sub * L4;
      const int t2 = 2 * t0 + 4 * q4;
      for (int orb = wave8 * RPI; 2 * orb < nrow; orb += NWV * RPI) {
        const int orow = orb + sub;
        if (2 * orow >= nrow || t2 >= 2 * d.Tout) continue;
        const int och = (rt0 * 16 >> 1) + orow;
        const bool has1 = 2 * orow + 1 < nrow;
        const float b0 = bias ? bias[rt0 * 16 + 2 * orow] : 0.f;
        const float b1 = (bias && has1) ? bias[rt0 * 16 + 2 * orow + 1] : 0.f;
        const float2 e0 = *reinterpret_cast<const float2*>(os + (2 * orow) * LDO + 2 * q4);
        const float2 e1 = has1 ? *reinterpret_cast<const float2*>(os + (2 * orow + 1) * LDO + 2 * q4) : make_float2(0.f, 0.f);
        f32x4 v = {e0.x + b0, e1.x + b1, e0.y + b0, e1.y + b1};
        const long oidx = ((long)b * (Cout >> 1) + och) * (2L * d.Tout) + t2;
        f32x4 rv = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
        if (d.res_mode == 1) rv = *reinterpret_cast<const f32x4*>(res + oidx);
        else if (d.res_mode == 2) rv = *reinterpret_cast<const f32x4*>(res + (long)b * 2 * d.Tout + t2);
        if (d.mul_mode) av = *reinterpret_cast<const f32x4*>(aux + oidx);
        f32x4* yp = reinterpret_cast<f32x4*>(y + oidx);
        const f32x4 y0 = d.accumulate ? *yp : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = v[e];
          if (d.res_mode) u += rv[e];
          u = nsc_apply_act(u, d.act);
          if (d.mul_mode) u *= nsc_act_grad_from_out(av[e], d.mul_mode);
          v[e] = u + y0[e];
        }
        if (has1) {
          NSC_CONV_ST4(yp, v);
        } else {                                   // (an odd channel count: the last output row has its even steps only)
          y[oidx] = v[0];
          y[oidx + 2] = v[2];
        }
      }
      return;
    }
    for (int orow = wave8; 2 * orow < nrow; orow += NWV) {
      const int och = (rt0 * 16 >> 1) + orow;
      const float b0 = bias ? bias[rt0 * 16 + 2 * orow] : 0.f;
      const float b1 = (bias && 2 * orow + 1 < nrow) ? bias[rt0 * 16 + 2 * orow + 1] : 0.f;
#pragma unroll
      for (int hb = 0; hb < 2 * TTc / 64; ++hb) {
        const int tl2 = hb * 64 + lane, par = tl2 & 1, tl = tl2 >> 1;
        const int t2 = 2 * t0 + tl2;
        if (t2 >= 2 * d.Tout || 2 * orow + par >= nrow) continue;
        float v = os[(2 * orow + par) * LDO + tl] + (par ? b1 : b0);
        const long oidx = ((long)b * (Cout >> 1) + och) * (2L * d.Tout) + t2;
        if (d.res_mode == 1) v += res[oidx];
        else if (d.res_mode == 2) v += res[(long)b * 2 * d.Tout + t2];
        v = nsc_apply_act(v, d.act);
        if (d.mul_mode) v *= nsc_act_grad_from_out(aux[oidx], d.mul_mode);
        if (d.accumulate) y[oidx] += v;
        else y[oidx] = v;
      }
    }
    return;
  }
  // Round 4: 16 bytes per lane where the rows allow it (T % 4 == 0, 16-byte aligned tensors): a wave instruction moves 64 / (TTc / 4) whole
  // rows - a quarter of the LDS reads and global stores of the dword form below (the k55 1 -> C conv spent 15 of its 23 us in launch +
  // this epilogue: 26 MB of row stores).
  if ((d.Tout & 3) == 0 && (LDO & 3) == 0 && ((uintptr_t)y & 15) == 0 && (!d.res_mode || ((uintptr_t)res & 15) == 0) &&
      (!d.mul_mode || ((uintptr_t)aux & 15) == 0)) {
    constexpr int L4 = TTc / 4, RPI = 64 / L4;
    static_assert(L4 <= 64 && 64 % L4 == 0, "a row's float4s divide the wave");
    const int sub = lane / L4, q4 = lane - sub * L4;
    const int t = t0 + 4 * q4;
    for (int rowb = wave8 * RPI; rowb < nrow; rowb += NWV * RPI) {
      const int row = rowb + sub;
      if (row >= nrow || t >= d.Tout) continue;
      const int ch = rt0 * 16 + row;
      const float bv = bias ? bias[ch] : 0.f;
      f32x4 v = *reinterpret_cast<const f32x4*>(os + row * LDO + 4 * q4);
      const long idx = ((long)b * Cout + ch) * d.Tout + t;
      f32x4 rv = {0.f, 0.f, 0.f, 0.f};
      if (d.res_mode == 1) rv = *reinterpret_cast<const f32x4*>(res + idx);
      else if (d.res_mode == 2) rv = *reinterpret_cast<const f32x4*>(res + (long)b * d.Tout + t);
      f32x4 av = {0.f, 0.f, 0.f, 0.f};
      if (d.mul_mode) av = *reinterpret_cast<const f32x4*>(aux + idx);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = v[e] + bv;
        if (d.res_mode) u += rv[e];
        u = nsc_apply_act(u, d.act);
        if (d.mul_mode) u *= nsc_act_grad_from_out(av[e], d.mul_mode);
        v[e] = u;
      }
      f32x4* yp = reinterpret_cast<f32x4*>(y + idx);
      if (d.accumulate) v += *yp;
      NSC_CONV_ST4(yp, v);
    }
    return;
  }
  for (int row = wave8; row < nrow; row += NWV) {
    const int ch = rt0 * 16 + row;
    const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
    for (int hb = 0; hb < TTc / 64; ++hb) {
      const int tl = hb * 64 + lane, t = t0 + tl;
      if (t >= d.Tout) continue;
      float v = os[row * LDO + tl] + bv;
      const long idx = ((long)b * Cout + ch) * d.Tout + t;
      if (d.res_mode == 1) v += res[idx];
      else if (d.res_mode == 2) v += res[(long)b * d.Tout + t];
      v = nsc_apply_act(v, d.act);
      if (d.mul_mode) v *= nsc_act_grad_from_out(aux[idx], d.mul_mode);
      const long oidx = d.out_mode == 1 ? ((long)b * (Cout >> 1) + (ch >> 1)) * (2L * d.Tout) + 2 * t + (ch & 1) : idx;
      if (d.accumulate) y[oidx] += v;
      else y[oidx] = v;
    }
  }
}

template <int RT, int NC, bool CIN1, int KS, int NPRE = 0>
__global__ __launch_bounds__(256 * KS) void conv1d_fwd_kernel(nsc_conv_desc d, const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ res,
                                                         const float* __restrict__ aux, float* __restrict__ y,
                                                         int ldx, int win) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (SGPR): selects, not branches
  const int wave = wave8 & 3, khalf = wave8 >> 2;
  const int l15 = lane & 15, kq = lane >> 4;
  const int TT = 4 * NC * 16;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int rt0 = blockIdx.z * RT;
  const int Cin4 = CIN1 ? 1 : ((d.Cin + 3) & ~3);
  const int Tin_virt = d.in_up ? 2 * d.Tin : d.Tin;
  const int skip = win >> 20;   // timing probe (NSC_CONV_SKIP): 1 staging, 2 MFMA loop, 4 epilogue
  win &= 0xfffff;

  float wa[NPRE > 0 ? NPRE : 1][RT];
  if constexpr (NPRE > 0) {
    static_assert(NPRE == 0 || KS == 1, "preloaded weights: one wave per (row tile group, column tile)");
#pragma unroll
    for (int u = 0; u < NPRE; ++u)
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int o = min((rt0 + r) * 16 + l15, d.Cout - 1);       // rows >= Cout: clamped, never stored
        if constexpr (CIN1) {
          const int tap = 4 * u + kq;                                // taps >= K (last k-step) need a real zero
          const float v = w[min(tap, d.K - 1) * d.Cout + o];
          wa[u][r] = (u == NPRE - 1 && tap >= d.K) ? 0.f : v;
        } else {
          const int ncq_ = Cin4 >> 2;
          const int tap = u / ncq_, ci = min(4 * (u - tap * ncq_) + kq, d.Cin - 1);   // ci >= Cin: zero rows of the x tile
          wa[u][r] = w[(min(tap, d.K - 1) * d.Cin + ci) * d.Cout + o];
        }
      }
  }
  // ---- stage x tile: xs[ci][j] = xin[b, ci, t0*stride - padL + j], zero outside (wave per row, lanes along time) ----
  if (!(skip & 1))
  nsc_stage_rows<4 * KS, NSC_CONV_U>(xs, ldx, Cin4, d.Cin, win, x + (long)b * d.Cin * d.Tin, d.Tin, t0 * d.stride - d.padL, Tin_virt,
                         d.in_up, wave8, lane);
  __syncthreads();

  f32x4 acc[RT][NC];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tcol0 = wave * NC * 16 + l15;  // this lane's first column (time offset inside the tile)
  const int Cout = d.Cout;

  if (skip & 2) {
  } else if constexpr (NPRE > 0) {
    const int ncq_ = CIN1 ? 1 : (Cin4 >> 2);
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      float bf[NC];
      if constexpr (CIN1) {
        const int tap = 4 * u + kq;
#pragma unroll
        for (int c = 0; c < NC; ++c) bf[c] = xs[(tcol0 + c * 16) * d.stride + tap * d.dil];
      } else {
        const int tap = u / ncq_, cq = u - tap * ncq_;
        const float* xrow = xs + kq * ldx + tcol0 * d.stride + (cq * 4 * ldx + tap * d.dil);
#pragma unroll
        for (int c = 0; c < NC; ++c) bf[c] = xrow[c * 16 * d.stride];
      }
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[u][r], bf[c], acc[r][c], 0, 0, 0);
    }
  } else if constexpr (CIN1) {
    // K order = tap; k-step s covers taps 4s..4s+3 (lane group kq).  xs has 3*dil zero slack at the end.
    // No control flow inside the MFMA loop (it would make hipcc shuttle the accumulators VGPR<->AGPR every step).
    const int nsteps = (d.K + 3) >> 2;
    float a_cur[RT], a_nxt[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int o = (rt0 + r) * 16 + l15;
      a_cur[r] = nsc_ldm(w, kq * Cout + o, kq < d.K && o < Cout);
    }
    for (int s = 0; s < nsteps; ++s) {
      const int tapn = 4 * (s + 1) + kq;
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int o = (rt0 + r) * 16 + l15;
        a_nxt[r] = nsc_ldm(w, tapn * Cout + o, tapn < d.K && o < Cout);
      }
      const int tap = 4 * s + kq;
      float bf[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) bf[c] = xs[(tcol0 + c * 16) * d.stride + tap * d.dil];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[r], bf[c], acc[r][c], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) a_cur[r] = a_nxt[r];
    }
  } else {
    // K order = (tap, ci) with ci fastest, ci padded to a multiple of 4 (pad rows of xs are zero).
    // Weight fragments: raw buffer loads (SRD built from wave-uniform values): per-lane byte offset is loop-invariant,
    // the k-step advances a SCALAR offset, and anything past the end of the weight array (padded k-steps) reads 0 from
    // the hardware bounds check - so the loop carries no masks, no selects and no per-load address arithmetic.
    // Fetched G k-steps ahead as a group so one L2 round trip is amortised over >= 32 MFMAs.  Rows o >= Cout (clamped)
    // and channels ci >= Cin (B rows are zero) need no masking: they only feed outputs that are never stored / add 0.
    constexpr int G = (RT * NC >= 14) ? 2 : ((RT * NC >= 7) ? 4 : ((RT * NC >= 4) ? 8 : 16));
    const int ncq = Cin4 >> 2;
    const int ntaps = (d.K - khalf + KS - 1) / KS;     // taps khalf, khalf + KS, ...
    const int nsteps = ntaps * ncq;
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(w), 0, d.K * d.Cin * Cout * 4, 0x00020000);
    int voff[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int o = (rt0 + r) * 16 + l15;
      voff[r] = (kq * Cout + (o < Cout ? o : Cout - 1)) * 4;
    }
    const int step_bytes = 4 * Cout * 4;          // one k-step = 4 input channels
    const int tap_bytes = d.Cin * Cout * 4;
    float an[G][RT];
    int tapp = khalf, cqp = 0;                     // prefetch cursor
    auto fetch = [&](float (&dst)[G][RT]) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int soff = __builtin_amdgcn_readfirstlane(tapp * tap_bytes + cqp * step_bytes);   // provably uniform: no waterfall
#pragma unroll
        for (int r = 0; r < RT; ++r)
          dst[u][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wsrd, voff[r], soff, 0));
        const bool wrapp = (cqp + 1 == ncq);
        cqp = wrapp ? 0 : cqp + 1;
        tapp += wrapp ? KS : 0;
      }
    };
    fetch(an);
    const int ngroups = (nsteps + G - 1) / G;      // padded steps multiply zero weights (out-of-range buffer reads)
    const int bbase = kq * ldx + tcol0 * d.stride; // per-lane part of the B (x tile) address
    int tap = khalf, cq = 0;
    for (int g = 0; g < ngroups; ++g) {
      float ac[G][RT];
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int r = 0; r < RT; ++r) ac[u][r] = an[u][r];
      fetch(an);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int tapc = tap < d.K ? tap : d.K - 1;
        const float* xrow = xs + bbase + __builtin_amdgcn_readfirstlane(cq * 4 * ldx + tapc * d.dil);
        float bf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) bf[c] = xrow[c * 16 * d.stride];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int c = 0; c < NC; ++c)
            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u][r], bf[c], acc[r][c], 0, 0, 0);
        const bool wrap = (cq + 1 == ncq);
        cq = wrap ? 0 : cq + 1;
        tap += wrap ? KS : 0;
      }
    }
  }

  if constexpr (KS == 2) {
    // partial sums of the odd-tap waves -> LDS (the x tile is dead by now) -> even-tap waves
    __syncthreads();
    float* red = xs + wave * (RT * NC * 256);
    if (khalf == 1) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) red[((r * NC + c) * 4 + reg) * 64 + lane] = acc[r][c][reg];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) acc[r][c][reg] += red[((r * NC + c) * 4 + reg) * 64 + lane];
    }
  }

  // ---- epilogue through LDS: the accumulators (D col = l15 -> time, row = 4*kq + reg -> channel) are transposed into an
  // [RT*16][TT] tile over the dead x tile, then written out row-wise: a wave handles whole 256-B rows, so y, the
  // residual and aux move as full cache lines (the D-fragment form wrote 64-B pieces and spent 16 of the 1x1 conv's
  // 34 us here) and the per-channel bias is a scalar.
  constexpr int TTc = 4 * NC * 16, LDO = TTc + 4;     // row stride == 4 (mod 32): fragment stores conflict-free
  __syncthreads();                                     // every wave is done reading the x tile / the split-K buffer
  float* os = xs;
  if (KS == 1 || khalf == 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int c = 0; c < NC; ++c) os[(r * 16 + kq * 4 + reg) * LDO + tcol0 + c * 16] = acc[r][c][reg];
  }
  __syncthreads();
  conv_store_rows<TTc, 4 * KS>(d, os, LDO, RT * 16, rt0, b, t0, wave8, lane, bias, res, aux, y);
}

// ------------------------------------------------------------------------------------------------
// The same implicit GEMM on v_mfma_f32_32x32x2_f32 for the convs with >= 96 output channels (the k9 stride-2 down-sampling
// conv and its polyphase data gradient: 12 GFLOP per step between them).  On this part the 32x32x2 shape sustains
// 147-150 TFLOP/s against 115-127 for 16x16x4 (tools/mfma_peak2.hip: it reads half the operands per flop), so the
// workgroup's ROWS = 32 NP + 16 output rows run as NP 32-row tiles plus one 16-row tile on the small shape (100 = 96 + 4
// channels: a fourth 32-row tile would be 7/8 padding).  Four waves x 32 time steps; KS = 2 splits the taps over eight
// waves as above; weights stream through raw buffer loads two k-steps (of 4 input channels) ahead.
// Fragment layouts (lane l): A32 row l % 32, k l / 32; B32 column l % 32, k l / 32; D32 register j <-> row
// 8 (j / 4) + 4 (l / 32) + j % 4, column l % 32.
// ------------------------------------------------------------------------------------------------
template <int NP, int KS>
__global__ __launch_bounds__(256 * KS) void conv1d_fwd_m32_kernel(nsc_conv_desc d, const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ res,
                                                                  const float* __restrict__ aux, float* __restrict__ y,
                                                                  int ldx, int win) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  constexpr int TT = 128, ROWS = 32 * NP + 16, NA = 2 * NP + 1, G = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3, khalf = wave8 >> 2;
  const int l31 = lane & 31, kh = lane >> 5, l15 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.y, t0 = blockIdx.x * TT, row0 = blockIdx.z * ROWS;
  const int Cin4 = (d.Cin + 3) & ~3, Cout = d.Cout;
  const int Tin_virt = d.in_up ? 2 * d.Tin : d.Tin;
  if (!d.in_up && Cin4 <= 104 && ldx <= 320 && KS == 2) {
    // Round 4: the whole x tile in flight at once (13 rows x 5 column blocks of dword buffer loads per lane, zeros outside the frame
    // from the bounds check), then the LDS stores: ONE memory round trip.  nsc_stage_rows batches U rows per round trip - ten round
    // trips for the 100 x 263 tile of the stride-2 conv, 10.5 of its 76 us with nothing to overlap them.
    constexpr int NQ = 13, NJ = 5;
    const __amdgpu_buffer_rsrc_t sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0,
                                                                        (unsigned)((long)d.B * d.Cin * d.Tin * 4), 0x00020000);
    float v[NQ][NJ];
    const int u0 = t0 * d.stride - d.padL;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      const int j = jb * 64 + lane, u = u0 + j;
      const int vo = (j < win && u >= 0 && u < d.Tin) ? u * 4 : 0x7ffffff0;
      if (jb * 64 < ldx) {                       // wave-uniform
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = wave8 + 8 * q;
          if (8 * q < Cin4)                      // wave-uniform guard: only the row groups this shape has
            v[q][jb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                     sx, r < d.Cin ? vo : 0x7ffffff0, (b * d.Cin + min(r, d.Cin - 1)) * d.Tin * 4, 0));
        }
      }
    }
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      const int j = jb * 64 + lane;
      if (j < ldx) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = wave8 + 8 * q;
          if (r < Cin4) xs[r * ldx + j] = v[q][jb];
        }
      }
    }
  } else {
    nsc_stage_rows<4 * KS, NSC_CONV_U>(xs, ldx, Cin4, d.Cin, win, x + (long)b * d.Cin * d.Tin, d.Tin, t0 * d.stride - d.padL,
                                       Tin_virt, d.in_up, wave8, lane);
  }
  __syncthreads();

  f32x16 acc[NP];
  f32x4 acc16[2];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[p][j] = 0.f;
  acc16[0] = acc16[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // weights: k-step = (tap, 4 input channels) = two K2 steps of the 32-row tiles + one K4 step of the 16-row tile.  Rows
  // >= Cout are clamped (never stored); k rows >= Cin meet zero rows of the x tile; anything past the array reads 0.
  const int ncq = Cin4 >> 2;
  const int ntaps = (d.K - khalf + KS - 1) / KS;
  const int nsteps = ntaps * ncq;
  const __amdgpu_buffer_rsrc_t wsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, d.K * d.Cin * Cout * 4, 0x00020000);
  int voff[NA];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int o = min(row0 + 32 * p + l31, Cout - 1);
    voff[2 * p] = (kh * Cout + o) * 4;                 // K2 half 0: input channels ci0 + {0, 1}
    voff[2 * p + 1] = ((2 + kh) * Cout + o) * 4;       // K2 half 1: ci0 + {2, 3}
  }
  voff[NA - 1] = (kq * Cout + min(row0 + 32 * NP + l15, Cout - 1)) * 4;
  const int step_bytes = 4 * Cout * 4, tap_bytes = d.Cin * Cout * 4;
  float an[G][NA];
  int tapp = khalf, cqp = 0;
  auto fetch = [&](float (&dst)[G][NA]) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int soff = __builtin_amdgcn_readfirstlane(tapp * tap_bytes + cqp * step_bytes);
#pragma unroll
      for (int q = 0; q < NA; ++q)
        dst[u][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wsrd, voff[q], soff, 0));
      const bool wrapp = (cqp + 1 == ncq);
      cqp = wrapp ? 0 : cqp + 1;
      tapp += wrapp ? KS : 0;
    }
  };
  fetch(an);
  const int ngroups = (nsteps + G - 1) / G;            // padded steps multiply zero weights (out-of-range buffer reads)
  const int bb32 = kh * ldx + (wave * 32 + l31) * d.stride;
  const int bb16 = kq * ldx + (wave * 32 + l15) * d.stride;
  int tap = khalf, cq = 0;
  for (int g = 0; g < ngroups; ++g) {
    float ac[G][NA];
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int q = 0; q < NA; ++q) ac[u][q] = an[u][q];
    fetch(an);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int tapc = tap < d.K ? tap : d.K - 1;
      const float* xr = xs + __builtin_amdgcn_readfirstlane(cq * 4 * ldx + tapc * d.dil);
      const float b0 = xr[bb32], b1 = xr[bb32 + 2 * ldx];
      const float c0 = xr[bb16], c1 = xr[bb16 + 16 * d.stride];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][2 * p], b0, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u][2 * p + 1], b1, acc[p], 0, 0, 0);
      }
      acc16[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u][NA - 1], c0, acc16[0], 0, 0, 0);
      acc16[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u][NA - 1], c1, acc16[1], 0, 0, 0);
      const bool wrap = (cq + 1 == ncq);
      cq = wrap ? 0 : cq + 1;
      tap += wrap ? KS : 0;
    }
  }

  constexpr int NREG = 16 * NP + 8;
  if constexpr (KS == 2) {
    // partial sums of the odd-tap waves -> LDS (the x tile is dead by now) -> even-tap waves
    __syncthreads();
    float* red = xs + wave * (NREG * 64);
    if (khalf == 1) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < 16; ++j) red[(16 * p + j) * 64 + lane] = acc[p][j];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[(16 * NP + 4 * c + reg) * 64 + lane] = acc16[c][reg];
    }
    __syncthreads();
    if (khalf == 0) {
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[p][j] += red[(16 * p + j) * 64 + lane];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc16[c][reg] += red[(16 * NP + 4 * c + reg) * 64 + lane];
    }
  }
  // output tile [ROWS][TT] through LDS, then whole rows out
  constexpr int LDO = TT + 4;
  __syncthreads();
  float* os = xs;
  if (KS == 1 || khalf == 0) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int j = 0; j < 16; ++j) os[(32 * p + (j >> 2) * 8 + kh * 4 + (j & 3)) * LDO + wave * 32 + l31] = acc[p][j];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) os[(32 * NP + kq * 4 + reg) * LDO + wave * 32 + 16 * c + l15] = acc16[c][reg];
  }
  __syncthreads();
  conv_store_rows<TT, 4 * KS>(d, os, LDO, ROWS, row0 / 16, b, t0, wave8, lane, bias, res, aux, y);
}

template <int NP, int KS>
static int launch_fwd_m32(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                          const float* aux, float* y, hipStream_t st, bool* taken) {
  constexpr int TT = 128, ROWS = 32 * NP + 16, NREG = 16 * NP + 8;
  const int win = (TT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  int ldx;
  if (d->stride == 1) { ldx = win; while ((ldx & 31) != 16) ++ldx; }
  else ldx = win | 1;
  const int Cin4 = (d->Cin + 3) & ~3;
  size_t smem = (size_t)Cin4 * ldx * sizeof(float);
  if (KS == 2) smem = std::max(smem, (size_t)4 * NREG * 64 * sizeof(float));
  smem = std::max(smem, (size_t)ROWS * (TT + 4) * sizeof(float));
  *taken = smem <= 160 * 1024;
  if (!*taken) return NSC_OK;
  auto kern = conv1d_fwd_m32_kernel<NP, KS>;
  const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_fwd_m32: set smem attr: %s", hipGetErrorString(e));
  dim3 grid(nsc_cdiv(d->Tout, TT), d->B, nsc_cdiv(d->Cout, ROWS));
  hipLaunchKernelGGL(kern, grid, dim3(256 * KS), smem, st, *d, x, w, bias, res, aux, y, ldx, win);
  NSC_CHECK_LAUNCH("conv1d_fwd_m32");
  return NSC_OK;
}

static int round_ldx_fwd(int win, int stride) {
  // conflict-free ds_read_b32 of the B fragment: lanes 0-15 read one row, 16-31 the next row.
  if (stride == 1) {
    int l = win;
    while ((l & 31) != 16) ++l;
    return l;
  }
  return win | 1;
}

template <int RT, int NC, bool CIN1, int KS, int NPRE = 0>
static int launch_fwd_ks(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                         const float* aux, float* y, hipStream_t st) {
  const int TT = 4 * NC * 16;
  int win = (TT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  if (CIN1) win += 3 * d->dil;  // slack so taps K..K+2 of the last k-step read zeros
  const int ldx = round_ldx_fwd(win, d->stride);
  const int Cin4 = CIN1 ? 1 : ((d->Cin + 3) & ~3);
  size_t smem = (size_t)Cin4 * ldx * sizeof(float);
  if (KS == 2) smem = std::max(smem, (size_t)4 * RT * NC * 256 * sizeof(float));
  smem = std::max(smem, (size_t)RT * 16 * (TT + 4) * sizeof(float));      // output tile of the row-wise epilogue
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "conv1d_fwd: x tile %zu B exceeds LDS", smem);
  auto kern = conv1d_fwd_kernel<RT, NC, CIN1, KS, NPRE>;
  if (smem > 64 * 1024) {
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_fwd: set smem attr: %s", hipGetErrorString(e));
  }
  const int nrt = nsc_cdiv(d->Cout, 16);
  dim3 grid(nsc_cdiv(d->Tout, TT), d->B, nsc_cdiv(nrt, RT));
  static const int skip = NSC_PROBE_INT("NSC_CONV_SKIP", 0);
  hipLaunchKernelGGL(kern, grid, dim3(256 * KS), smem, st, *d, x, w, bias, res, aux, y, ldx, win | (skip << 20));
  NSC_CHECK_LAUNCH("conv1d_fwd");
  return NSC_OK;
}

template <int RT, int NC, bool CIN1>
static int launch_fwd(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                      const float* aux, float* y, hipStream_t st) {
  static const bool no_pre = NSC_PROBE_SET("NSC_CONV_NOPRE");   // A/B switch for profiling
  if constexpr (!CIN1) {
    if (d->K >= 2) return launch_fwd_ks<RT, NC, CIN1, 2>(d, x, w, bias, res, aux, y, st);
    if constexpr (RT == 7) {
      if (!no_pre && d->K == 1 && ((d->Cin + 3) >> 2) == 25) return launch_fwd_ks<RT, NC, CIN1, 1, 25>(d, x, w, bias, res, aux, y, st);
    }
  } else if constexpr (RT == 7 || RT == 4) {
    if (!no_pre && d->K > 52 && d->K <= 56) return launch_fwd_ks<RT, NC, CIN1, 1, 14>(d, x, w, bias, res, aux, y, st);
  }
  return launch_fwd_ks<RT, NC, CIN1, 1>(d, x, w, bias, res, aux, y, st);
}

template <int NC, bool CIN1>
static int dispatch_rt(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                       const float* aux, float* y, hipStream_t st) {
  const int nrt = nsc_cdiv(d->Cout, 16);
  switch (nrt) {
    case 1: return launch_fwd<1, NC, CIN1>(d, x, w, bias, res, aux, y, st);
    case 2: return launch_fwd<2, NC, CIN1>(d, x, w, bias, res, aux, y, st);
    case 3: return launch_fwd<3, NC, CIN1>(d, x, w, bias, res, aux, y, st);
    case 4: return launch_fwd<4, NC, CIN1>(d, x, w, bias, res, aux, y, st);
    case 5: case 6: case 7: return launch_fwd<7, NC, CIN1>(d, x, w, bias, res, aux, y, st);
    default: return launch_fwd<4, NC, CIN1>(d, x, w, bias, res, aux, y, st);  // grid.z walks groups of 4 row tiles
  }
}

static int check_desc(const nsc_conv_desc* d, const char* who) {
  NSC_REQUIRE(d, NSC_ERR_BAD_ARG, "%s: null desc", who);
  NSC_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->Tin > 0 && d->Tout > 0 && d->K > 0 && d->dil > 0 &&
                  d->stride > 0 && d->padL >= 0,
              NSC_ERR_BAD_ARG, "%s: non-positive size in desc (B=%d Cin=%d Cout=%d Tin=%d Tout=%d K=%d)", who, d->B,
              d->Cin, d->Cout, d->Tin, d->Tout, d->K);
  NSC_REQUIRE(d->act >= 0 && d->act <= 2 && d->res_mode >= 0 && d->res_mode <= 2 && d->mul_mode >= 0 &&
                  d->mul_mode <= 2 && d->out_mode >= 0 && d->out_mode <= 1,
              NSC_ERR_BAD_ARG, "%s: bad mode field", who);
  NSC_REQUIRE(d->stride <= 2, NSC_ERR_UNSUPPORTED, "%s: stride %d > 2 unsupported", who, d->stride);
  NSC_REQUIRE(!(d->out_mode == 1 && (d->Cout & 1)), NSC_ERR_BAD_ARG, "%s: shuffle needs even Cout", who);
  NSC_REQUIRE(!(d->in_up && d->stride != 1), NSC_ERR_UNSUPPORTED, "%s: in_up needs stride 1", who);
  return NSC_OK;
}

extern "C" int nsc_conv1d_fwd(const nsc_conv_desc* d, const float* x, const float* w, const float* bias,
                              const float* res, const float* aux, float* y, void* stream) {
  int rc = check_desc(d, "nsc_conv1d_fwd");
  if (rc) return rc;
  NSC_REQUIRE(x && w && y, NSC_ERR_BAD_ARG, "nsc_conv1d_fwd: null x/w/y");
  NSC_REQUIRE(!(d->res_mode && !res), NSC_ERR_BAD_ARG, "nsc_conv1d_fwd: res_mode set but res null");
  NSC_REQUIRE(!(d->mul_mode && !aux), NSC_ERR_BAD_ARG, "nsc_conv1d_fwd: mul_mode set but aux null");
  hipStream_t st = (hipStream_t)stream;
  // NC=2 (128-step time tiles) when the x tile stays small enough for >= 2 workgroups per CU.
  const bool cin1 = d->Cin == 1;
  const int Cin4 = cin1 ? 1 : ((d->Cin + 3) & ~3);
  const long smem2 = (long)Cin4 * ((127L) * d->stride + (d->K - 1) * d->dil + 40) * 4;
  static const int force_nc = NSC_PROBE_INT("NSC_CONV_NC", 0);   // profiling switch
  bool nc2 = smem2 <= 72 * 1024 && d->Tout >= 128;
  if (force_nc) nc2 = force_nc == 2;
  static const bool no_m32 = NSC_PROBE_SET("NSC_CONV_NO_M32");   // A/B switch for profiling
  if (!cin1 && !no_m32 && d->K >= 2 && d->Cout >= 96 && d->Tout >= 128) {
    bool taken = false;
    rc = launch_fwd_m32<3, 2>(d, x, w, bias, res, aux, y, st, &taken);
    if (rc || taken) return rc;
  }
  if (cin1) return nc2 ? dispatch_rt<2, true>(d, x, w, bias, res, aux, y, st)
                       : dispatch_rt<1, true>(d, x, w, bias, res, aux, y, st);
  return nc2 ? dispatch_rt<2, false>(d, x, w, bias, res, aux, y, st)
             : dispatch_rt<1, false>(d, x, w, bias, res, aux, y, st);
}

// ------------------------------------------------------------------------------------------------
// Cout == 1 (k55 C->1): VALU dot products; 128 outputs per workgroup, channels split over 2 thread halves
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1d_cout1_kernel(nsc_conv_desc d, const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ res,
                                                           const float* __restrict__ aux, float* __restrict__ y,
                                                           int ldx) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* xs = sm;                       // [Cin][ldx]
  float* ws = sm + d.Cin * ldx;         // [K*Cin]  (w[k,ci,0])
  float* part = ws + d.K * d.Cin;       // [128]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * 128;
  const int Tin_virt = d.in_up ? 2 * d.Tin : d.Tin;
  const int u0 = t0 * d.stride - d.padL;
  const float* xb = x + (long)b * d.Cin * d.Tin;
  nsc_stage_rows(xs, ldx, d.Cin, d.Cin, ldx, xb, d.Tin, u0, Tin_virt, d.in_up, tid >> 6, tid & 63);
  for (int e = tid; e < d.K * d.Cin; e += 256) ws[e] = w[e];
  __syncthreads();
  const int tl = tid & 127, half = tid >> 7;
  const int chalf = (d.Cin + 1) >> 1;
  const int c0 = half * chalf, c1 = min(d.Cin, c0 + chalf);
  float acc = 0.f;
  for (int ci = c0; ci < c1; ++ci) {
    const float* xr = xs + ci * ldx + tl * d.stride;
    for (int k = 0; k < d.K; ++k) acc = fmaf(xr[k * d.dil], ws[k * d.Cin + ci], acc);
  }
  if (half == 1) part[tl] = acc;
  __syncthreads();
  if (half == 0) {
    const int t = t0 + tl;
    if (t < d.Tout) {
      float v = acc + part[tl] + (bias ? bias[0] : 0.f);
      const long idx = (long)b * d.Tout + t;
      if (d.res_mode) v += res[idx];
      v = nsc_apply_act(v, d.act);
      if (d.mul_mode) v *= nsc_act_grad_from_out(aux[idx], d.mul_mode);
      if (d.accumulate) y[idx] += v;
      else y[idx] = v;
    }
  }
}

// Register-tiled form for the k55 convs of the model (stride 1, dilation 1): the v1 kernel above issues two LDS reads
// per FMA and runs at ~5 TF/s.  Here a lane owns R consecutive outputs and slides a (R+K-1)-sample window held in
// registers over the taps, so a channel costs ~(R+K)/4 + K/4 16-byte LDS reads for R*K FMAs (VALU-bound); the weights
// are read as wave-uniform (broadcast) float4s from a [Cin][K] transposed copy.  8 waves split the channels and meet
// in LDS.
template <int K, int R>
__global__ __launch_bounds__(512) void conv1d_cout1_v2_kernel(nsc_conv_desc d, const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res,
                                                              const float* __restrict__ aux, float* __restrict__ y,
                                                              int ldx, nsc_cout1_chain ch, nsc_cout1_quant qz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64 * R, KP = (K + 3) & ~3, NWIN = (R + KP - 1 + 3) & ~3;
  static_assert(NWIN % R == 0, "window is read in R-float vectors");
  float* xs = sm;                       // [Cin][ldx]   ldx >= TT + NWIN - R, multiple of 4, zero-filled past the tile
  float* ws = xs + d.Cin * ldx;         // [Cin][KP]    ws[c][k] = w[k][c], zero for k >= K
  float* part = ws + d.Cin * KP;        // [8][TT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  // (Round 4 tried consuming the rows one at a time in load order - a wave computes exactly the channels it loads, so a row needs no
  // workgroup barrier and the later rows' latency could run under the earlier rows' FMAs: the 13-fold unrolled body measured the same
  // at B = 128 (17.4 / 11.4 / 18.7 us against 16.1 / 11.4 / 19.1) and 40 % SLOWER at B = 4096 (328 vs 236 us); not kept.)
  {
    // whole x tile in flight at once: wave w owns rows w, w+8, ...; raw buffer loads, out-of-frame columns -> 0 by the
    // hardware bounds check (a batched-by-8 staging loop spent 5 memory round trips here: 25 of the kernel's 50 us)
    constexpr int NQ = 13, NJ = (TT + NWIN - R + 63) / 64;      // up to 104 channels
    const __amdgpu_buffer_rsrc_t sx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0,
                                                                        (unsigned)((long)d.B * d.Cin * d.Tin * 4), 0x00020000);
    float v[NQ][NJ];
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      const int j = jb * 64 + lane;
      const int u = t0 - d.padL + j;
      const int vo = (j < TT + K - 1 && u >= 0 && u < d.Tin) ? u * 4 : 0x7ffffff0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int r = min(wave + 8 * q, d.Cin - 1);
        v[q][jb] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sx, vo, (b * d.Cin + r) * d.Tin * 4, 0));
      }
    }
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      const int j = jb * 64 + lane;
      if (j < ldx) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const int r = wave + 8 * q;
          if (r < d.Cin) xs[r * ldx + j] = v[q][jb];
        }
      }
    }
  }
  for (int e = tid; e < d.Cin * KP; e += 512) {
    const int c = e / KP, k = e - c * KP;
    ws[e] = k < K ? w[k * d.Cin + c] : 0.f;
  }
  __syncthreads();
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.f;
  for (int c = wave; c < d.Cin; c += 8) {
    // the window starts at float R*lane of the row: read it as R-float vectors so consecutive lanes hit consecutive
    // banks (scalar reads at a 4-float lane stride were 8-way bank conflicts: 44 of this kernel's 51 us)
    typedef float vecR __attribute__((ext_vector_type(R)));
    const vecR* xr = reinterpret_cast<const vecR*>(xs) + (c * (ldx / R) + lane);
    const float* wr = ws + c * KP;
    float xw[NWIN];
#pragma unroll
    for (int i = 0; i < NWIN / R; ++i) {
      const vecR t = xr[i];
#pragma unroll
      for (int e = 0; e < R; ++e) xw[i * R + e] = t[e];
    }
#pragma unroll
    for (int k4 = 0; k4 < KP / 4; ++k4) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * k4);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (4 * k4 + kk < K) {
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r] = fmaf(xw[4 * k4 + kk + r], wv[kk], acc[r]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) part[wave * TT + R * lane + r] = acc[r];
  __syncthreads();
  if (tid < TT) {
    const int t = t0 + tid;
    if (t < d.Tout) {
      float v = bias ? bias[0] : 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v += part[q * TT + tid];
      const long idx = (long)b * d.Tout + t;
      if (d.res_mode) v += res[idx];
      v = nsc_apply_act(v, d.act);
      if (d.mul_mode) v *= nsc_act_grad_from_out(aux[idx], d.mul_mode);
      if (d.accumulate) v += y[idx];
      y[idx] = v;
      // elementwise chain on the [B,1,T] result (nsc_conv1d_cout1_fwd_chain): the cascade step behind a codec's output conv, the
      // running sum / next codec's output gradient behind the first conv's data gradient - each was a launch of its own
      if (ch.out2) {
        const float o2 = ch.p_in ? fmaf(ch.pb, v, ch.pa * ch.p_in[idx]) : ch.pb * v;
        ch.out2[idx] = o2;
        if (ch.out3) ch.out3[idx] = fmaf(ch.qa, ch.q_in[idx], ch.qb * o2);
      }
      if (qz.qcode) xs[tid] = v;                 // the tile's codes for the quantizer stage below (the x tile is dead)
    } else if (qz.qcode) {
      xs[tid] = 0.f;
    }
  }
  // ---- fused soft-to-hard quantizer of the training step (nn_core_operator.py:140-164 on the code this conv just produced; p is
  // not materialised: quan_loss partials and the soft histogram only, as nsc_quantize_fwd with p_out = null).  It was a launch of its
  // own behind every encoder (8-10 us of latency for 32 k codes).  32 bins: 4 lanes per code, 8 bins per lane, 128 codes per pass.
  if (qz.qcode) {
    float* qsh = part + 8 * TT;                  // [32 + 8] behind the partial-sum tile (dynamic LDS: the kernel asks for all 160 KB)
    if (tid < 40) qsh[tid] = 0.f;
    __syncthreads();
    constexpr int LPC = 4, ITER = 2;
    const int gl = tid & 3;
    const float alpha = qz.alpha[0];
    float bv[ITER][4], hacc[ITER][4];
    bool ok[ITER][4];
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bv[i][j] = qz.bins[(i * LPC + gl) * 4 + j];
        ok[i][j] = true;
        hacc[i][j] = 0.f;
      }
    float qacc = 0.f;
#pragma unroll
    for (int ps = 0; ps < TT / 128; ++ps) {
      const int ci = ps * 128 + (tid >> 2), t = t0 + ci;
      const bool live = t < d.Tout;
      const float c = xs[ci];
      float dist[ITER][4], p[ITER][4];
      softmax_bins<LPC, ITER>(c, alpha, bv, ok, dist, p);
      float q;
      if (qz.soft) {
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < ITER; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) s_ = fmaf(p[i][j], bv[i][j], s_);
        q = grp_sum<LPC>(s_);
      } else {
        float best = -1.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < ITER; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = (i * LPC + gl) * 4 + j;
            if (p[i][j] > best) { best = p[i][j]; bi = k; }
          }
        bi = grp_argmax<LPC>(best, bi);
        q = (c != c) ? c : qz.bins[min(bi, 31)];
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < ITER; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hacc[i][j] += p[i][j];
            qacc += __builtin_amdgcn_sqrtf(p[i][j] + QEPS);
          }
        if (gl == 0) qz.qcode[(long)b * d.Tout + t] = (1.f - qz.is_quan_on) * c + qz.is_quan_on * q;
      }
    }
    // histogram: lanes with the same gl (bins) across the wave's 16 codes, then the 8 waves through LDS, one atomic per bin and workgroup
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float h = hacc[i][j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) h += __shfl_xor(h, o, 64);
        if (lane < 4) atomicAdd(&qsh[(i * LPC + gl) * 4 + j], h);
      }
    qacc = wave_sum(qacc);
    if (lane == 0) qsh[32 + wave] = qacc;
    __syncthreads();
    if (qz.hist && tid < 32) atomicAdd(qz.hist + tid, qsh[tid]);
    if (qz.quan && tid == 0) {
      float s_ = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) s_ += qsh[32 + w8];
      atomicAdd(qz.quan + b, s_ / (float)d.Tout);       // (a frame's tiles add up: the caller zeroes quan)
    }
  }
}

template <int K, int R>
static int launch_cout1_v2(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res,
                           const float* aux, float* y, hipStream_t st, const nsc_cout1_chain& ch, const nsc_cout1_quant& qz) {
  constexpr int TT = 64 * R, KP = (K + 3) & ~3, NWIN = (R + KP - 1 + 3) & ~3;
  const int ldx = (TT + NWIN - R + 3) & ~3;
  const size_t smem = ((size_t)d->Cin * ldx + (size_t)d->Cin * KP + 8 * TT + 64) * sizeof(float);
  if (smem > 160 * 1024) return 1;     // does not fit: caller falls back to v1
  auto kern = conv1d_cout1_v2_kernel<K, R>;
  const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
  NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_cout1_v2: set smem attr: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(kern, dim3(nsc_cdiv(d->Tout, TT), d->B), dim3(512), smem, st, *d, x, w, bias, res, aux, y, ldx, ch, qz);
  NSC_CHECK_LAUNCH("conv1d_cout1_v2");
  return NSC_OK;
}

// elementwise chain behind a Cout = 1 conv, for the shapes the register-tiled kernel does not serve
__global__ void cout1_chain_kernel(const float* __restrict__ v, nsc_cout1_chain ch, long n) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const float o2 = ch.p_in ? fmaf(ch.pb, v[e], ch.pa * ch.p_in[e]) : ch.pb * v[e];
    ch.out2[e] = o2;
    if (ch.out3) ch.out3[e] = fmaf(ch.qa, ch.q_in[e], ch.qb * o2);
  }
}

static int cout1_fwd_impl(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, const float* res, const float* aux,
                          float* y, const nsc_cout1_chain* chain, const nsc_cout1_quant* quant, void* stream) {
  int rc = check_desc(d, "nsc_conv1d_cout1_fwd");
  if (rc) return rc;
  nsc_cout1_chain ch;
  memset(&ch, 0, sizeof(ch));
  if (chain) ch = *chain;
  nsc_cout1_quant qz;
  memset(&qz, 0, sizeof(qz));
  if (quant) {
    qz = *quant;
    NSC_REQUIRE(qz.qcode && qz.alpha && qz.bins, NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd_quant: null qcode / alpha / bins");
    // only the register-tiled kernel carries the quantizer stage (32 bins); anything else: NSC_ERR_UNSUPPORTED, the caller launches
    // the conv and nsc_quantize_fwd separately
    NSC_REQUIRE(qz.nb == 32 && d->K == 55 && d->dil == 1 && d->stride == 1 && !d->in_up && d->Cin >= 8 && d->Cin <= 104 && !d->accumulate,
                NSC_ERR_UNSUPPORTED, "nsc_conv1d_cout1_fwd_quant: needs 32 bins and the k55 C -> 1 shapes (C in 8..104)");
  }
  NSC_REQUIRE(!ch.out3 || (ch.out2 && ch.q_in), NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd_chain: out3 needs out2 and q_in");
  NSC_REQUIRE(d->Cout == 1 && d->out_mode == 0, NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd: needs Cout == 1, plain store");
  NSC_REQUIRE(x && w && y, NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd: null x/w/y");
  NSC_REQUIRE(!(d->res_mode && !res) && !(d->mul_mode && !aux), NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd: null res/aux");
  static const bool v1_only = NSC_PROBE_SET("NSC_COUT1_V1");   // A/B switch for profiling
  if (!v1_only && d->K == 55 && d->dil == 1 && d->stride == 1 && !d->in_up && d->Cin >= 8 && d->Cin <= 104) {
    // R outputs per lane: 4 when that still gives every CU a workgroup, else 2
    // (R = 4 already at half a chip's worth of workgroups: the launch is latency, not throughput - B = 128, T = 256: 128 workgroups of
    // 256 steps take 22.2 us, 256 workgroups of 128 steps 24.5)
    static const int force_r = NSC_PROBE_INT("NSC_COUT1_R", 0);      // (probes build: 2 | 4 forces the outputs per lane - round 6, GPU-sharing study)
    const bool r4 = force_r ? force_r == 4 : (long)d->B * nsc_cdiv(d->Tout, 256) >= 128;
    const int rc2 = r4 ? launch_cout1_v2<55, 4>(d, x, w, bias, res, aux, y, (hipStream_t)stream, ch, qz)
                       : launch_cout1_v2<55, 2>(d, x, w, bias, res, aux, y, (hipStream_t)stream, ch, qz);
    if (rc2 <= 0) return rc2;          // launched (0) or failed (<0); 1 = tile does not fit LDS -> v1 below
  }
  NSC_REQUIRE(!qz.qcode, NSC_ERR_UNSUPPORTED, "nsc_conv1d_cout1_fwd_quant: the tile does not fit the register-tiled kernel");
  int ldx = 127 * d->stride + (d->K - 1) * d->dil + 1;
  ldx |= 1;
  const size_t smem = ((size_t)d->Cin * ldx + (size_t)d->K * d->Cin + 128) * sizeof(float);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "conv1d_cout1: tile %zu B exceeds LDS", smem);
  if (smem > 64 * 1024) {
    const hipError_t e = NSC_SMEM_ATTR(conv1d_cout1_kernel, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_cout1: set smem attr: %s", hipGetErrorString(e));
  }
  dim3 grid(nsc_cdiv(d->Tout, 128), d->B);
  hipLaunchKernelGGL(conv1d_cout1_kernel, grid, dim3(256), smem, (hipStream_t)stream, *d, x, w, bias, res, aux, y, ldx);
  NSC_CHECK_LAUNCH("conv1d_cout1");
  if (ch.out2) {                       // (other shapes: the chain as a launch of its own)
    const long n = (long)d->B * d->Tout;
    hipLaunchKernelGGL(cout1_chain_kernel, dim3(std::min<long>(4096, nsc_cdiv(n, 256))), dim3(256), 0, (hipStream_t)stream, y, ch, n);
    NSC_CHECK_LAUNCH("cout1_chain");
  }
  return NSC_OK;
}

extern "C" int nsc_conv1d_cout1_fwd_chain(const nsc_conv_desc* d, const float* x, const float* w, const float* bias,
                                          const float* res, const float* aux, float* y, const nsc_cout1_chain* chain,
                                          void* stream) {
  return cout1_fwd_impl(d, x, w, bias, res, aux, y, chain, nullptr, stream);
}
extern "C" int nsc_conv1d_cout1_fwd_quant(const nsc_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                                          const nsc_cout1_quant* quant, void* stream) {
  NSC_REQUIRE(quant, NSC_ERR_BAD_ARG, "nsc_conv1d_cout1_fwd_quant: null quant");
  return cout1_fwd_impl(d, x, w, bias, nullptr, nullptr, y, nullptr, quant, stream);
}
extern "C" int nsc_conv1d_cout1_fwd(const nsc_conv_desc* d, const float* x, const float* w, const float* bias,
                                    const float* res, const float* aux, float* y, void* stream) {
  return cout1_fwd_impl(d, x, w, bias, res, aux, y, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient as a GEMM with the (b, t) axis as the reduction:
//     dW[(tap,ci), o] = sum_{b,t} x[b, ci, t*stride + tap*dil - padL] * dz[b, o, t]      (+ a row of ones -> db)
// Rows (tap,ci) are split over grid.y (128*RT rows per workgroup: 8 waves x RT row tiles), the (b,t) reduction over
// grid.x ("K-splits"); every workgroup keeps its RT x CT accumulator tiles in registers across all of its 64-step
// chunks.  Both operand tiles are staged with 16 row loads in flight per wave (the first version issued one dz row per
// round trip and spent ~80 % of its time there).  Flush: with a workspace each K-split STORES its partial dW into a
// private slab and conv_slab_reduce_kernel sums the slabs (192 K-splits x 90k float atomics on the same addresses took
// 2/3 of the stride-2 conv's 283 us); without one, float atomics straight into dw/db (few K-splits).
// ------------------------------------------------------------------------------------------------
// bx / gx: this workgroup's K-split index and the number of K-splits; by: its row group
#ifndef NSC_CW_UNROLL
#define NSC_CW_UNROLL 4   // k-steps unrolled together (2: +3-7 % launch time; 8: same as 4)
#endif
template <int RT, int CT>
__device__ __forceinline__ void conv_wgrad_body(const nsc_conv_desc& d, const float* __restrict__ x,
                                                const float* __restrict__ dz, float* __restrict__ dw,
                                                float* __restrict__ db, float* __restrict__ slab, long slab_stride,
                                                int flip, int ldx, int win, int nchunk_t, int bx, int gx, int by) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int TT = 64, LDZ = 66;  // 66 % 32 == 2: conflict-free B-fragment reads
  float* xs = sm;                          // [Cin + 2][ldx] : rows Cin = zeros, Cin+1 = ones
  float* dzs = sm + (d.Cin + 2) * ldx;     // [CT*16][LDZ]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR): selects, not branches
  const int l15 = lane & 15, kq = lane >> 4;
  const int nW = d.K * d.Cin;
  const int nrows = nW + (db ? 1 : 0);
  const int Tin_virt = d.in_up ? 2 * d.Tin : d.Tin;
  const int rt0 = (by * 8 + wave) * RT;   // first row tile of this wave

  // per-lane A-row offsets (row = rowtile*16 + l15)
  int rowoff[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const int kk = (rt0 + r) * 16 + l15;
    if (kk < nW) {
      const int tap = kk / d.Cin, ci = kk - tap * d.Cin;
      rowoff[r] = ci * ldx + tap * d.dil;
    } else if (kk < nrows) {
      rowoff[r] = (d.Cin + 1) * ldx;  // ones row -> bias gradient
    } else {
      rowoff[r] = d.Cin * ldx;        // zero row
    }
  }
  f32x4 acc[RT][CT];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // constant rows
  for (int j = tid; j < ldx; j += 512) {
    xs[d.Cin * ldx + j] = 0.f;
    xs[(d.Cin + 1) * ldx + j] = 1.f;
  }
  const bool busy = rt0 * 16 < nrows;   // waves past the last row tile only help staging
  const int nchunks = d.B * nchunk_t;
  auto mfma_chunk = [&]() {
#pragma unroll (RT >= 4 ? 2 : NSC_CW_UNROLL)
    for (int tt = 0; tt < TT / 4; ++tt) {
      const int tloc = 4 * tt + kq;
      float af[RT], bf[CT];
#pragma unroll
      for (int r = 0; r < RT; ++r) af[r] = xs[rowoff[r] + tloc * d.stride];
#pragma unroll
      for (int c = 0; c < CT; ++c) bf[c] = dzs[(c * 16 + l15) * LDZ + tloc];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], bf[c], acc[r][c], 0, 0, 0);
    }
  };
  // Register prefetch of the next chunk's operand tiles (raw buffer loads; out-of-frame columns come back as 0 from the
  // bounds check) while this chunk's MFMAs run: staging and MFMA time were equal (~7 us per chunk of the stride-2 conv),
  // serial.  Shapes outside the register budget (Cin > 104, window > 192 columns, zero-upsampled input) stage in place.
  constexpr int NQX = 13, NH = 3, NQZ = 2 * CT;
  if (!d.in_up && d.Cin <= 8 * NQX && win <= 64 * NH) {
    const __amdgpu_buffer_rsrc_t sx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (unsigned)((long)d.B * d.Cin * d.Tin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sz =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz), 0, (unsigned)((long)d.B * d.Cout * d.Tout * 4), 0x00020000);
    float px[NQX][NH], pz[NQZ];
    const int OOB = 0x7ffffff0;
    auto load_chunk = [&](int chunk) {
      const int cc = __builtin_amdgcn_readfirstlane(chunk < nchunks ? chunk : bx);
      const int b = cc / nchunk_t, t0 = (cc - b * nchunk_t) * TT;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int j = h * 64 + lane;
        const int u = t0 * d.stride - d.padL + j;
        const int vo = (j < win && u >= 0 && u < d.Tin) ? u * 4 : OOB;
        if (h * 64 < win) {                         // wave-uniform guards: only the rows / column blocks this shape has
#pragma unroll
          for (int q = 0; q < NQX; ++q)
            if (8 * q < d.Cin)
              px[q][h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                       sx, vo, (b * d.Cin + min(wave + 8 * q, d.Cin - 1)) * d.Tin * 4, 0));
        }
      }
      const int vz = (t0 + lane < d.Tout) ? (t0 + lane) * 4 : OOB;
#pragma unroll
      for (int q = 0; q < NQZ; ++q)
        pz[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                              sz, vz, (b * d.Cout + min(wave + 8 * q, d.Cout - 1)) * d.Tout * 4, 0));
    };
    auto store_chunk = [&]() {
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int j = h * 64 + lane;
        if (j < ldx) {
#pragma unroll
          for (int q = 0; q < NQX; ++q) {
            const int r = wave + 8 * q;
            if (r < d.Cin) xs[r * ldx + j] = px[q][h];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < NQZ; ++q) {
        const int o = wave + 8 * q;
        dzs[o * LDZ + lane] = o < d.Cout ? pz[q] : 0.f;
      }
    };
    load_chunk(bx);
    for (int chunk = bx; chunk < nchunks; chunk += gx) {
      __syncthreads();  // previous chunk's reads done
      store_chunk();
      __syncthreads();
      load_chunk(chunk + gx);
      if (busy) mfma_chunk();
    }
  } else {
    for (int chunk = bx; chunk < nchunks; chunk += gx) {
      const int b = chunk / nchunk_t, tc = chunk - b * nchunk_t;
      const int t0 = tc * TT;
      __syncthreads();  // previous chunk's reads done
      nsc_stage_rows<8, 16>(xs, ldx, d.Cin, d.Cin, win, x + (long)b * d.Cin * d.Tin, d.Tin, t0 * d.stride - d.padL,
                            Tin_virt, d.in_up, wave, lane);
      nsc_stage_rows<8, 16>(dzs, LDZ, CT * 16, d.Cout, TT, dz + (long)b * d.Cout * d.Tout, d.Tout, t0, d.Tout, 0, wave,
                            lane, TT);
      __syncthreads();
      if (busy) mfma_chunk();
    }
  }
  if (!busy) return;
  // ---- flush: D col = l15 -> output channel, row = 4*kq + reg -> kk ----
  float* sl = slab ? slab + (long)bx * slab_stride : nullptr;
#pragma unroll
  for (int r = 0; r < RT; ++r) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int kk = (rt0 + r) * 16 + kq * 4 + reg;
      if (kk >= nrows) continue;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int o = c * 16 + l15;
        if (o >= d.Cout) continue;
        const float v = acc[r][c][reg];
        if (sl) {
          sl[(long)kk * d.Cout + o] = v;     // rows in kernel order; the bias row is row nW; flip applied by the reduce
        } else if (kk < nW) {
          int kko = kk;
          if (flip) {
            const int tap = kk / d.Cin, ci = kk - tap * d.Cin;
            kko = (d.K - 1 - tap) * d.Cin + ci;
          }
          atomicAdd(dw + (long)kko * d.Cout + o, v);
        } else {
          atomicAdd(db + o, v);
        }
      }
    }
  }
}

template <int RT, int CT>
__global__ __launch_bounds__(512) void conv1d_wgrad_kernel(nsc_conv_desc d, const float* __restrict__ x,
                                                           const float* __restrict__ dz, float* __restrict__ dw,
                                                           float* __restrict__ db, float* __restrict__ slab,
                                                           long slab_stride, int flip, int ldx, int win, int nchunk_t) {
  conv_wgrad_body<RT, CT>(d, x, dz, dw, db, slab, slab_stride, flip, ldx, win, nchunk_t, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---- batched form: the weight gradients of up to NSC_CW_MAXJ convs (same RT/CT class) in ONE launch; every job always
// flushes to its own slabs, summed by one conv_slab_reduce_batch_kernel launch.  The engine defers the per-conv weight
// gradients to the end of the backward pass: alone, each is a ~30 us launch + a reduce launch for ~5 us of MFMA work.
#define NSC_CW_MAXJ 10
struct ConvWgradJobDev {
  nsc_conv_desc d;
  const float *x, *dz;
  float *dw, *db;            // db != null marks "has bias row"
  float* slab;
  long slab_stride;
  int flip, ldx, win, nchunk_t, gx, gy, wg0;
};
struct ConvWgradBatch {
  ConvWgradJobDev j[NSC_CW_MAXJ];
  int njobs;
};
template <int RT, int CT>
__global__ __launch_bounds__(512) void conv1d_wgrad_batch_kernel(ConvWgradBatch t) {
  const int w = blockIdx.x;
  int j = 0;
#pragma unroll
  for (int q = 1; q < NSC_CW_MAXJ; ++q)
    if (q < t.njobs && w >= t.j[q].wg0) j = q;
  // copy the selected job with wave-uniform selects (a dynamically indexed kernarg array is spilled to scratch)
  ConvWgradJobDev jb = t.j[0];
#pragma unroll
  for (int q = 1; q < NSC_CW_MAXJ; ++q)
    if (q == j) jb = t.j[q];
  const int wl = w - jb.wg0;
  conv_wgrad_body<RT, CT>(jb.d, jb.x, jb.dz, jb.dw, jb.db, jb.slab, jb.slab_stride, jb.flip, jb.ldx, jb.win, jb.nchunk_t,
                          wl % jb.gx, jb.gx, wl / jb.gx);
}

struct ConvReduceJob {
  const float* slab;
  long stride;
  float *dw, *db;
  int nslabs, K, Cin, Cout, n, flip;
};
#define NSC_CR_MAXJ 16   // jobs per reduce launch: ONE reduce serves all kernel classes of a step (round 4; it ran once per class)
struct ConvReduceBatch {
  ConvReduceJob j[NSC_CR_MAXJ];
};
// 1024 threads = 64 consecutive elements x 16 slab groups (one wave each; its loads are whole 256-B lines): the slabs of an
// element are summed by 16 waves in parallel (a single thread walking all of them is a ~100-load dependent chain),
// the 16 partial sums meet in LDS in a fixed order: results do not depend on timing.
__global__ __launch_bounds__(1024) void conv_slab_reduce_batch_kernel(ConvReduceBatch t) {
  __shared__ float part[16][64];
  ConvReduceJob jb = t.j[0];
#pragma unroll
  for (int q = 1; q < NSC_CR_MAXJ; ++q)
    if (q == (int)blockIdx.y) jb = t.j[q];
  const int nW = jb.K * jb.Cin * jb.Cout;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int per = (jb.nslabs + 15) >> 4;
  const int w0 = grp * per, w1 = min(jb.nslabs, w0 + per);
  for (int i0 = blockIdx.x * 64; i0 < jb.n; i0 += gridDim.x * 64) {     // block-uniform trip count
    const int i = i0 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < jb.n) {
      int w = w0;
      for (; w + 3 < w1; w += 4) {
        s0 += jb.slab[(long)w * jb.stride + i];
        s1 += jb.slab[(long)(w + 1) * jb.stride + i];
        s2 += jb.slab[(long)(w + 2) * jb.stride + i];
        s3 += jb.slab[(long)(w + 3) * jb.stride + i];
      }
      for (; w < w1; ++w) s0 += jb.slab[(long)w * jb.stride + i];
    }
    part[grp][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && i < jb.n) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += part[g][lane];
      if (i < nW) {
        int io = i;
        if (jb.flip) {
          const int kk = i / jb.Cout, o = i - kk * jb.Cout;
          const int tap = kk / jb.Cin, ci = kk - tap * jb.Cin;
          io = ((jb.K - 1 - tap) * jb.Cin + ci) * jb.Cout + o;
        }
        jb.dw[io] += v;          // one adder per element
      } else {
        jb.db[i - nW] += v;
      }
    }
    __syncthreads();
  }
}

// dw / db += sum over the K-split slabs (blockIdx.y splits the slabs so enough loads are in flight)
__global__ void conv_slab_reduce_kernel(const float* __restrict__ slab, long stride, int nslabs, float* __restrict__ dw,
                                        float* __restrict__ db, int K, int Cin, int Cout, int n, int flip) {
  const int per = (nslabs + gridDim.y - 1) / gridDim.y;
  const int w0 = blockIdx.y * per, w1 = min(nslabs, w0 + per);
  const int nW = K * Cin * Cout;
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
    const float v = (s0 + s1) + (s2 + s3);
    if (i < nW) {
      int io = i;
      if (flip) {
        const int kk = i / Cout, o = i - kk * Cout;
        const int tap = kk / Cin, ci = kk - tap * Cin;
        io = ((K - 1 - tap) * Cin + ci) * Cout + o;
      }
      atomicAdd(dw + io, v);
    } else {
      atomicAdd(db + (i - nW), v);
    }
  }
}

struct WgradPlan { int gx, gy, ldx, win, nchunk_t; size_t smem; long slab_stride; };

template <int RT>
static WgradPlan wgrad_plan(const nsc_conv_desc* d, int CT, bool bias, bool use_slab) {
  WgradPlan p;
  const int TT = 64;
  p.win = (TT - 1) * d->stride + (d->K - 1) * d->dil + 1;
  p.ldx = p.win;
  while ((p.ldx & 31) != 2) ++p.ldx;  // A-fragment rows (consecutive ci) land on distinct even banks, kq on odd
  p.smem = ((size_t)(d->Cin + 2) * p.ldx + (size_t)CT * 16 * 66) * sizeof(float);
  const int nrows = d->K * d->Cin + (bias ? 1 : 0);
  const int nrt = nsc_cdiv(nrows, 16);
  p.gy = nsc_cdiv(nrt, 8 * RT);
  p.nchunk_t = nsc_cdiv(d->Tout, TT);
  const int nchunks = d->B * p.nchunk_t;
  // K-splits: with slabs ~one workgroup per CU (256 in all) and >= 2 chunks each (the slab traffic grows with gx);
  // with atomics keep the number of same-address adders small.
  static const int gx_base = NSC_PROBE_INT("NSC_WGRAD_GX", 256);   // tuning probe
  static const int gx_dbl = NSC_PROBE_INT("NSC_WGRAD_GX2", 1);
  int gx = use_slab ? gx_base / p.gy : 64 / p.gy;
  if (p.smem <= 76 * 1024 && use_slab && gx_dbl) gx *= 2;      // two workgroups fit a CU
  if (gx > nchunks / 2) gx = nchunks / 2;
  if (gx < 1) gx = 1;
  p.gx = gx;
  p.slab_stride = ((long)nrows * d->Cout + 63) & ~63L;
  return p;
}

static bool wgrad_big(const nsc_conv_desc* d, bool bias) { return nsc_cdiv(d->K * d->Cin + (bias ? 1 : 0), 16) > 24; }
// Row tiles per wave.  Every workgroup stages the whole x and dz tile of its chunk but owns only 8 RT of dW's row tiles, so a
// tall dW pays the staging once per row group: the stride-2 k9 100 -> 100 conv (57 row tiles) at RT = 2 staged every chunk
// four times, 24 of its 100 us (probe builds without the loads / the LDS stores / the MFMA loop: profiles/r03d).  RT = 4 (112
// accumulator registers, 7-column-tile class only) halves that.
static int wgrad_rt_class(const nsc_conv_desc* d, bool bias, int ct) {
  static const bool no_rt4 = NSC_PROBE_SET("NSC_WGRAD_NO_RT4");   // A/B switch for profiling
  const int nrt = nsc_cdiv(d->K * d->Cin + (bias ? 1 : 0), 16);
  if (nrt > 48 && ct == 7 && !no_rt4) return 4;
  return nrt > 24 ? 2 : 1;
}

template <int RT, int CT>
static int launch_wgrad(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db, int flip,
                        float* ws, long ws_floats, hipStream_t st) {
  WgradPlan p = wgrad_plan<RT>(d, CT, db != nullptr, ws != nullptr);
  if (ws && (p.gx < 4 || (long)p.gx * p.slab_stride > ws_floats)) {   // not worth a slab / does not fit: atomics
    ws = nullptr;
    p = wgrad_plan<RT>(d, CT, db != nullptr, false);
  }
  NSC_REQUIRE(p.smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "conv1d_wgrad: tiles %zu B exceed LDS", p.smem);
  auto kern = conv1d_wgrad_kernel<RT, CT>;
  if (p.smem > 64 * 1024) {
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_wgrad: set smem attr: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(p.gx, p.gy), dim3(512), p.smem, st, *d, x, dz, dw, db, ws, p.slab_stride, flip, p.ldx,
                     p.win, p.nchunk_t);
  NSC_CHECK_LAUNCH("conv1d_wgrad");
  if (ws) {
    const int n = (d->K * d->Cin + (db ? 1 : 0)) * d->Cout;
    const int gy = p.gx >= 64 ? 8 : (p.gx >= 16 ? 4 : 1);
    hipLaunchKernelGGL(conv_slab_reduce_kernel, dim3(nsc_cdiv(n, 256), gy), dim3(256), 0, st, ws, p.slab_stride, p.gx, dw,
                       db, d->K, d->Cin, d->Cout, n, flip);
    NSC_CHECK_LAUNCH("conv_slab_reduce");
  }
  return NSC_OK;
}

template <int CT>
static int dispatch_wgrad_rt(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db, int flip,
                             float* ws, long ws_floats, hipStream_t st) {
  const int rt = wgrad_rt_class(d, db != nullptr, CT);
  if (rt == 1) return launch_wgrad<1, CT>(d, x, dz, dw, db, flip, ws, ws_floats, st);
  if constexpr (CT == 7) {
    if (rt == 4) return launch_wgrad<4, CT>(d, x, dz, dw, db, flip, ws, ws_floats, st);   // very tall dW (stride-2 k9 100->100)
  }
  return launch_wgrad<2, CT>(d, x, dz, dw, db, flip, ws, ws_floats, st);
}

extern "C" long nsc_conv1d_wgrad_workspace(const nsc_conv_desc* d) {
  if (!d || d->K <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
  int ct = nsc_cdiv(d->Cout, 16);
  if (ct > 4) ct = 7;
  const int rt = wgrad_rt_class(d, true, ct);
  const WgradPlan p = rt == 4 ? wgrad_plan<4>(d, ct, true, true) : (rt == 2 ? wgrad_plan<2>(d, ct, true, true) : wgrad_plan<1>(d, ct, true, true));
  return (long)p.gx * p.slab_stride;
}

extern "C" int nsc_conv1d_wgrad_ws(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db,
                                   int flip_taps, float* workspace, long workspace_floats, void* stream) {
  int rc = check_desc(d, "nsc_conv1d_wgrad");
  if (rc) return rc;
  NSC_REQUIRE(x && dz && dw, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad: null x/dz/dw");
  NSC_REQUIRE(d->Cout <= 112, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad: Cout %d > 112", d->Cout);
  hipStream_t st = (hipStream_t)stream;
  const int nct = nsc_cdiv(d->Cout, 16);
  switch (nct) {
    case 1: return dispatch_wgrad_rt<1>(d, x, dz, dw, db, flip_taps, workspace, workspace_floats, st);
    case 2: return dispatch_wgrad_rt<2>(d, x, dz, dw, db, flip_taps, workspace, workspace_floats, st);
    case 3: return dispatch_wgrad_rt<3>(d, x, dz, dw, db, flip_taps, workspace, workspace_floats, st);
    case 4: return dispatch_wgrad_rt<4>(d, x, dz, dw, db, flip_taps, workspace, workspace_floats, st);
    default: return dispatch_wgrad_rt<7>(d, x, dz, dw, db, flip_taps, workspace, workspace_floats, st);
  }
}

extern "C" int nsc_conv1d_wgrad(const nsc_conv_desc* d, const float* x, const float* dz, float* dw, float* db,
                                int flip_taps, void* stream) {
  return nsc_conv1d_wgrad_ws(d, x, dz, dw, db, flip_taps, nullptr, 0, stream);
}

// ---- weight gradient of a ONE-INPUT-CHANNEL conv (the k55 1 -> C input convs of the codecs), round 4 ----
//   dW[tap][o] = sum_{b,t} x[b, t + tap - padL] dz[b, o, t]      db[o] = sum_{b,t} dz[b, o, t]
// In conv_wgrad_body the rows of the GEMM are (tap, ci): with one input channel that is 56 rows - four of the eight waves have a row
// tile, each workgroup sees 2 chunks, and the launch is all prologue, flush and slab traffic (30 us for 0.7 GFLOP and 26 MB).  Here the
// roles are swapped: rows = OUTPUT CHANNELS (one 16-row tile per wave: 7 of 8 busy at C = 100), columns = taps (4 tiles; column K is
// the bias: B = 1), reduction over time.  The k index of a step is mapped to time as t = t_blk + 16 kq + ks, so a lane's A operands of
// 16 consecutive k-steps are 16 consecutive floats of its dz row: four 16-byte loads straight into registers, no LDS for dz at all;
// B is the x window in LDS, consecutive lanes consecutive addresses.  A workgroup walks chunks of 256 steps; partial dW^T leaves to
// the job's slab in the layout conv_slab_reduce_batch_kernel expects ([tap][o], bias row behind the taps).
#define NSC_CW1_MAXJ 8
struct Cin1WgradJob {
  const float *x, *dz;
  float* slab;
  long slab_stride;
  int B, Cout, T, K, padL, nrows, gx, wg0;
};
struct Cin1WgradBatch {
  Cin1WgradJob j[NSC_CW1_MAXJ];
  int njobs;
};
__global__ __launch_bounds__(512) void conv1d_wgrad_cin1_kernel(Cin1WgradBatch tb) {
  __shared__ float xs[320];
  int jq = 0;
#pragma unroll
  for (int q = 1; q < NSC_CW1_MAXJ; ++q)
    if (q < tb.njobs && (int)blockIdx.x >= tb.j[q].wg0) jq = q;
  Cin1WgradJob jb = tb.j[0];
#pragma unroll
  for (int q = 1; q < NSC_CW1_MAXJ; ++q)
    if (q == jq) jb = tb.j[q];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int wl = blockIdx.x - jb.wg0;
  const int nchunk_t = jb.T >> 8, nchunks = jb.B * nchunk_t;
  const bool busy = wave * 16 < jb.Cout;
  const int o = wave * 16 + l15;
  const __amdgpu_buffer_rsrc_t sz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(jb.dz), 0, (unsigned)((long)jb.B * jb.Cout * jb.T * 4), 0x00020000);
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // column tile 3 holds taps 48..63: the real ones, the bias column (tap == K: B = 1) and zero columns - a per-lane constant
  const int tap3 = 48 + l15;
  const float one3 = (tap3 == jb.K && jb.nrows > jb.K) ? 1.f : 0.f;
  const bool real3 = tap3 < jb.K;
  // this lane's dz row, 16 consecutive steps per 64-step block: a chunk's 16 loads; the NEXT chunk's go out before this chunk's MFMAs
  // (register double buffer) and the barriers order LDS traffic only, so the loads stay in flight across them
  f32x4 a4[2][4][4];
  auto load_a = [&](int chunk, int buf) {
    const int cc = chunk < nchunks ? chunk : wl;             // past the end: a harmless re-read
    const int b = cc / nchunk_t, t0 = (cc - b * nchunk_t) << 8;
    const int vo = (busy && o < jb.Cout) ? ((b * jb.Cout + o) * jb.T + t0 + 16 * kq) * 4 : 0x7ffffff0;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        a4[buf][blk][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sz, vo, (64 * blk + 4 * i) * 4, 0));
  };
  auto run_chunk = [&](int chunk, auto buf_c) {
    constexpr int BUF = decltype(buf_c)::value;
    const int b = chunk / nchunk_t, t0 = (chunk - b * nchunk_t) << 8;
    nsc_lds_barrier();                             // the previous chunk's window reads are done
    if (tid < 320) {
      const int u = t0 - jb.padL + tid;
      xs[tid] = (u >= 0 && u < jb.T) ? jb.x[(long)b * jb.T + u] : 0.f;
    }
    nsc_lds_barrier();
    load_a(chunk + jb.gx, BUF ^ 1);
    if (busy) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const float* xb = xs + 64 * blk + 16 * kq + l15;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const float a = a4[BUF][blk][ks >> 2][ks & 3];
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[ks + 16 * c], acc[c], 0, 0, 0);
          const float b3 = real3 ? xb[ks + 48] : one3;
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b3, acc[3], 0, 0, 0);
        }
      }
    }
  };
  load_a(wl, 0);
  for (int chunk = wl; chunk < nchunks; chunk += 2 * jb.gx) {
    run_chunk(chunk, std::integral_constant<int, 0>{});
    if (chunk + jb.gx < nchunks) run_chunk(chunk + jb.gx, std::integral_constant<int, 1>{});
  }
  if (!busy) return;
  float* sl = jb.slab + (long)wl * jb.slab_stride;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int tap = 16 * c + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int oo = wave * 16 + kq * 4 + r;
      if (oo < jb.Cout && tap < jb.nrows) sl[(long)tap * jb.Cout + oo] = acc[c][r];
    }
  }
}
static bool cw_is_cin1(const nsc_conv_wgrad_job& jb) {
  static const bool off = NSC_PROBE_SET("NSC_CW_NO_CIN1");        // A/B switch for profiling
  const nsc_conv_desc& d = jb.d;
  return !off && d.Cin == 1 && d.stride == 1 && d.dil == 1 && !d.in_up && d.Tin == d.Tout && (d.Tout & 255) == 0 && d.K + (jb.db ? 1 : 0) <= 64 &&
         d.K >= 48 && d.Cout <= 112 && d.padL <= 63 && (long)d.B * d.Cout * d.Tout * 4 < (1L << 31) && (((uintptr_t)jb.dz) & 15) == 0;
}
static long cw_cin1_stride(const nsc_conv_wgrad_job& jb) { return (((long)(jb.d.K + (jb.db ? 1 : 0)) * jb.d.Cout) + 63) & ~63L; }
// The one-input-channel jobs of a call, in launches of up to NSC_CW1_MAXJ jobs; a launch's workgroups (256: one per CU - at 896 the
// headline step's six jobs took 66 us in two launches, every workgroup a single chunk) are shared out by the jobs' chunk counts.
// idx[i] = job index, gx[i] = its workgroups (= slabs), launch_of[i] = launch number.  Returns the number of such jobs.
static int cw_cin1_plan(const nsc_conv_wgrad_job* jobs, int njobs, int* idx, int* gx, int* launch_of) {
  int n = 0;
  for (int j = 0; j < njobs && n < 256; ++j)
    if (cw_is_cin1(jobs[j])) idx[n++] = j;
  for (int lo = 0; lo < n; lo += NSC_CW1_MAXJ) {
    const int hi = std::min(n, lo + NSC_CW1_MAXJ);
    long tot = 0;
    for (int i = lo; i < hi; ++i) tot += (long)jobs[idx[i]].d.B * (jobs[idx[i]].d.Tout >> 8);
    for (int i = lo; i < hi; ++i) {
      const long ch = (long)jobs[idx[i]].d.B * (jobs[idx[i]].d.Tout >> 8);
      gx[i] = (int)std::max(1L, std::min(ch, (256 * ch + tot / 2) / tot));
      launch_of[i] = lo / NSC_CW1_MAXJ;
    }
  }
  return n;
}

// ---- host side of the batched form ----
struct CwPlan { int rt, ct, gy, nchunks; long stride; size_t smem; double weight; WgradPlan p; };

static CwPlan cw_plan(const nsc_conv_wgrad_job& jb) {
  CwPlan c;
  const nsc_conv_desc* d = &jb.d;
  const bool bias = jb.db != nullptr;
  c.ct = nsc_cdiv(d->Cout, 16);
  if (c.ct > 4) c.ct = 7;
  c.rt = wgrad_rt_class(d, bias, c.ct);
  c.p = c.rt == 4 ? wgrad_plan<4>(d, c.ct, bias, true) : (c.rt == 2 ? wgrad_plan<2>(d, c.ct, bias, true) : wgrad_plan<1>(d, c.ct, bias, true));
  c.gy = c.p.gy;
  c.nchunks = d->B * c.p.nchunk_t;
  c.stride = c.p.slab_stride;
  c.smem = c.p.smem;
  c.weight = (double)c.nchunks * c.gy * (d->Cin * nsc_cdiv(c.p.win, 64) + c.ct * 16 + 4.0 * c.rt * c.ct);
  return c;
}

// K-splits per job of one class: a budget of workgroups shared in proportion to the jobs' weights
static void cw_split(const CwPlan* c, int n, int* gx) {
  size_t smem = 0;
  double tot = 0;
  for (int q = 0; q < n; ++q) { smem = std::max(smem, c[q].smem); tot += c[q].weight; }
  static const int budget_probe = NSC_PROBE_INT("NSC_CW_BUDGET", 0);      // tuning probe (probes build only)
  // one workgroup per CU for every class: with two per CU (512) the small-LDS classes ran 2 chunks per workgroup and their prologue,
  // flush and slab traffic outweighed the overlap (class <1,7> of the headline step: 73.3 us at 512, 67.4 at 256, 84.4 at 384 / 1024)
  const int budget = budget_probe ? budget_probe : 256;
  (void)smem;
  for (int q = 0; q < n; ++q) {
    int g = (int)(budget * c[q].weight / tot / c[q].gy + 0.5);
    g = std::min(g, std::max(1, c[q].nchunks / 2));
    gx[q] = std::max(1, g);
  }
}

// the slab reductions of a step: collected class by class, launched once (or every NSC_CR_MAXJ jobs)
struct CwReduceQueue {
  ConvReduceBatch r;
  int n = 0, nmax = 0;
  long off = 0;              // floats of the workspace handed out so far: the slabs of ALL classes coexist until the reduce
  int flush(hipStream_t st) {
    if (n == 0) return NSC_OK;
    hipLaunchKernelGGL(conv_slab_reduce_batch_kernel, dim3(std::min(512, nsc_cdiv(nmax, 64)), n), dim3(1024), 0, st, r);
    NSC_CHECK_LAUNCH("conv_slab_reduce_batch");
    n = 0; nmax = 0;
    return NSC_OK;
  }
};

template <int RT, int CT>
static int launch_cw_class(const nsc_conv_wgrad_job* jobs, const int* idx, const CwPlan* cp, int n, float* workspace,
                           long workspace_floats, hipStream_t st, CwReduceQueue& rq) {
  ConvWgradBatch t;
  memset(&t, 0, sizeof(t));
  int gx[NSC_CW_MAXJ];
  cw_split(cp, n, gx);
  if (rq.n + n > NSC_CR_MAXJ) {
    int rc = rq.flush(st);
    if (rc) return rc;
  }
  ConvReduceBatch& r = rq.r;
  long off = rq.off;
  int wg = 0, nmax = rq.nmax;
  size_t smem = 0;
  for (int q = 0; q < n; ++q) {
    const nsc_conv_wgrad_job& jb = jobs[idx[q]];
    ConvWgradJobDev& dv = t.j[q];
    dv.d = jb.d; dv.x = jb.x; dv.dz = jb.dz; dv.dw = jb.dw; dv.db = jb.db;
    dv.slab = workspace + off; dv.slab_stride = cp[q].stride;
    dv.flip = jb.flip_taps; dv.ldx = cp[q].p.ldx; dv.win = cp[q].p.win; dv.nchunk_t = cp[q].p.nchunk_t;
    dv.gx = gx[q]; dv.gy = cp[q].gy; dv.wg0 = wg;
    ConvReduceJob& rj = r.j[rq.n + q];
    rj.slab = dv.slab; rj.stride = dv.slab_stride; rj.dw = jb.dw; rj.db = jb.db; rj.nslabs = gx[q];
    rj.K = jb.d.K; rj.Cin = jb.d.Cin; rj.Cout = jb.d.Cout; rj.flip = jb.flip_taps;
    rj.n = (jb.d.K * jb.d.Cin + (jb.db ? 1 : 0)) * jb.d.Cout;
    nmax = std::max(nmax, rj.n);
    off += (long)gx[q] * cp[q].stride;
    wg += gx[q] * cp[q].gy;
    smem = std::max(smem, cp[q].smem);
  }
  NSC_REQUIRE(off <= workspace_floats, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_batch: workspace %ld floats < %ld", workspace_floats, off);
  NSC_REQUIRE(smem <= 160 * 1024, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_batch: tiles %zu B exceed LDS", smem);
  t.njobs = n;
  auto kern = conv1d_wgrad_batch_kernel<RT, CT>;
  if (smem > 64 * 1024) {
    const hipError_t e = NSC_SMEM_ATTR(kern, 160 * 1024);
    NSC_REQUIRE(e == hipSuccess, NSC_ERR_LAUNCH, "conv1d_wgrad_batch: set smem attr: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3(wg), dim3(512), smem, st, t);
  NSC_CHECK_LAUNCH("conv1d_wgrad_batch");
  rq.n += n;
  rq.nmax = nmax;
  rq.off = off;
  return NSC_OK;
}

static int cw_dispatch(int rt, int ct, const nsc_conv_wgrad_job* jobs, const int* idx, const CwPlan* cp, int n, float* ws,
                       long wsf, hipStream_t st, CwReduceQueue& rq) {
#define CW(RT_, CT_) if (rt == RT_ && ct == CT_) return launch_cw_class<RT_, CT_>(jobs, idx, cp, n, ws, wsf, st, rq)
  CW(1, 1); CW(1, 2); CW(1, 3); CW(1, 4); CW(1, 7); CW(2, 1); CW(2, 2); CW(2, 3); CW(2, 4); CW(2, 7); CW(4, 7);
#undef CW
  nsc_set_error("nsc_conv1d_wgrad_batch: no kernel for class (%d, %d)", rt, ct);
  return NSC_ERR_UNSUPPORTED;
}

// visits the jobs class by class ((RT, CT) template instance), at most NSC_CW_MAXJ per launch; f(rt, ct, idx, plans, n)
template <typename F>
static int cw_for_each_class(const nsc_conv_wgrad_job* jobs, int njobs, F f) {
  bool done[256] = {false};
  NSC_REQUIRE(njobs <= 256, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_batch: more than 256 jobs");
  for (int a = 0; a < njobs; ++a) {
    if (done[a]) continue;
    const CwPlan pa = cw_plan(jobs[a]);
    int idx[NSC_CW_MAXJ], n = 0;
    CwPlan cp[NSC_CW_MAXJ];
    for (int b = a; b < njobs && n < NSC_CW_MAXJ; ++b) {
      if (done[b]) continue;
      const CwPlan pb = cw_plan(jobs[b]);
      if (pb.rt == pa.rt && pb.ct == pa.ct) { idx[n] = b; cp[n] = pb; ++n; done[b] = true; }
    }
    int rc = f(pa.rt, pa.ct, idx, cp, n);
    if (rc) return rc;
  }
  return NSC_OK;
}

extern "C" long nsc_conv1d_wgrad_batch_workspace(const nsc_conv_wgrad_job* jobs, int njobs) {
  if (!jobs || njobs <= 0) return 0;
  long need = 0;       // the slabs of every class live until the one reduce at the end: the SUM over the classes
  nsc_conv_wgrad_job rest[256];
  int nrest = 0;
  {
    int idx1[256], gx1[256], l1[256];
    const int n1 = cw_cin1_plan(jobs, njobs, idx1, gx1, l1);
    for (int i = 0; i < n1; ++i) need += (long)gx1[i] * cw_cin1_stride(jobs[idx1[i]]);
  }
  for (int j = 0; j < njobs && j < 256; ++j)
    if (!cw_is_cin1(jobs[j])) rest[nrest++] = jobs[j];
  jobs = rest;
  njobs = nrest;
  if (njobs == 0) return need;
  cw_for_each_class(jobs, njobs, [&](int, int, const int*, const CwPlan* cp, int n) {
    int gx[NSC_CW_MAXJ];
    cw_split(cp, n, gx);
    for (int q = 0; q < n; ++q) need += (long)gx[q] * cp[q].stride;
    return 0;
  });
  return need;
}

extern "C" int nsc_conv1d_wgrad_batch(const nsc_conv_wgrad_job* jobs, int njobs, float* workspace, long workspace_floats,
                                      void* stream) {
  NSC_REQUIRE(jobs && njobs > 0 && workspace, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_batch: bad arguments");
  for (int j = 0; j < njobs; ++j) {
    int rc = check_desc(&jobs[j].d, "nsc_conv1d_wgrad_batch");
    if (rc) return rc;
    NSC_REQUIRE(jobs[j].x && jobs[j].dz && jobs[j].dw, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_batch: job %d: null x/dz/dw", j);
    NSC_REQUIRE(jobs[j].d.Cout <= 112, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_batch: job %d: Cout %d > 112", j, jobs[j].d.Cout);
  }
  hipStream_t st = (hipStream_t)stream;
  CwReduceQueue rq;
  memset(&rq.r, 0, sizeof(rq.r));
  NSC_REQUIRE(njobs <= 256, NSC_ERR_UNSUPPORTED, "nsc_conv1d_wgrad_batch: more than 256 jobs");
  // the one-input-channel k55 convs first: their own kernel (conv1d_wgrad_cin1_kernel), up to NSC_CW1_MAXJ jobs per launch
  nsc_conv_wgrad_job rest[256];
  int nrest = 0;
  {
    Cin1WgradBatch tb;
    memset(&tb, 0, sizeof(tb));
    int wg = 0;
    auto flush1 = [&]() -> int {
      if (tb.njobs == 0) return NSC_OK;
      hipLaunchKernelGGL(conv1d_wgrad_cin1_kernel, dim3(wg), dim3(512), 0, st, tb);
      NSC_CHECK_LAUNCH("conv1d_wgrad_cin1");
      memset(&tb, 0, sizeof(tb));
      wg = 0;
      return NSC_OK;
    };
    int idx1[256], gx1[256], l1[256];
    const int n1 = cw_cin1_plan(jobs, njobs, idx1, gx1, l1);
    for (int j = 0; j < njobs; ++j)
      if (!cw_is_cin1(jobs[j])) rest[nrest++] = jobs[j];
    for (int i = 0; i < n1; ++i) {
      const nsc_conv_wgrad_job& jb = jobs[idx1[i]];
      if (i > 0 && l1[i] != l1[i - 1]) { int rc1 = flush1(); if (rc1) return rc1; }
      if (rq.n == NSC_CR_MAXJ) { int rc1 = flush1(); if (rc1) return rc1; rc1 = rq.flush(st); if (rc1) return rc1; }
      const int gx = gx1[i], nrows = jb.d.K + (jb.db ? 1 : 0);
      const long stride = cw_cin1_stride(jb);
      NSC_REQUIRE(rq.off + (long)gx * stride <= workspace_floats, NSC_ERR_BAD_ARG, "nsc_conv1d_wgrad_batch: workspace %ld floats too small", workspace_floats);
      Cin1WgradJob& q = tb.j[tb.njobs++];
      q.x = jb.x; q.dz = jb.dz; q.slab = workspace + rq.off; q.slab_stride = stride; q.B = jb.d.B; q.Cout = jb.d.Cout; q.T = jb.d.Tout; q.K = jb.d.K;
      q.padL = jb.d.padL; q.nrows = nrows; q.gx = gx; q.wg0 = wg;
      wg += gx;
      ConvReduceJob& rj = rq.r.j[rq.n++];
      rj.slab = q.slab; rj.stride = stride; rj.dw = jb.dw; rj.db = jb.db; rj.nslabs = gx; rj.K = jb.d.K; rj.Cin = 1; rj.Cout = jb.d.Cout;
      rj.flip = jb.flip_taps; rj.n = nrows * jb.d.Cout;
      rq.nmax = std::max(rq.nmax, rj.n);
      rq.off += (long)gx * stride;
    }
    int rc1 = flush1();
    if (rc1) return rc1;
  }
  jobs = rest;
  njobs = nrest;
  if (njobs == 0) return rq.flush(st);
  int rc = cw_for_each_class(jobs, njobs, [&](int rt, int ct, const int* idx, const CwPlan* cp, int n) {
    return cw_dispatch(rt, ct, jobs, idx, cp, n, workspace, workspace_floats, st, rq);
  });
  if (rc) return rc;
  return rq.flush(st);
}

// ------------------------------------------------------------------------------------------------
// wt[K-1-k, o, i] = w[k, i, o]
// ------------------------------------------------------------------------------------------------
__global__ void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int K, int Cin,
                                             int Cout) {
  const int n = K * Cin * Cout;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    // e indexes wt [K][Cout][Cin]
    const int i = e % Cin;
    const int o = (e / Cin) % Cout;
    const int kt = e / (Cin * Cout);
    wt[e] = w[((long)(K - 1 - kt) * Cin + i) * Cout + o];
  }
}
// the four kernels of a gated block in ONE launch: wt = wt1 [1][20][Cin] | wtl [15][20][20] | wtr [15][20][20] | wt9 [9][C][20]
struct Flip4 { const float* w[4]; int K[4], ci[4], co[4], end[4]; };
__global__ void weight_flip_transpose4_kernel(Flip4 f, float* __restrict__ wt) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < f.end[3]; e += gridDim.x * blockDim.x) {
    const int q = e < f.end[0] ? 0 : (e < f.end[1] ? 1 : (e < f.end[2] ? 2 : 3));
    const int l = e - (q ? f.end[q - 1] : 0);
    const int Cin = f.ci[q], Cout = f.co[q];
    const int i = l % Cin, o = (l / Cin) % Cout, kt = l / (Cin * Cout);
    wt[e] = f.w[q][((long)(f.K[q] - 1 - kt) * Cin + i) * Cout + o];
  }
}
extern "C" int nsc_gated_block_flip_weights(const float* w1, const float* wl, const float* wr, const float* w9, float* wt, int C,
                                            int Cin, int narrow, int k9, void* stream) {
  NSC_REQUIRE(w1 && wl && wr && w9 && wt && C > 0 && Cin > 0 && narrow > 0 && k9 > 0, NSC_ERR_BAD_ARG, "nsc_gated_block_flip_weights: bad args");
  Flip4 f;
  f.w[0] = w1; f.K[0] = 1; f.ci[0] = Cin; f.co[0] = narrow;
  f.w[1] = wl; f.K[1] = 15; f.ci[1] = narrow; f.co[1] = narrow;
  f.w[2] = wr; f.K[2] = 15; f.ci[2] = narrow; f.co[2] = narrow;
  f.w[3] = w9; f.K[3] = k9; f.ci[3] = narrow; f.co[3] = C;
  int end = 0;
  for (int q = 0; q < 4; ++q) { end += f.K[q] * f.ci[q] * f.co[q]; f.end[q] = end; }
  hipLaunchKernelGGL(weight_flip_transpose4_kernel, dim3(std::min(1024, nsc_cdiv(end, 256))), dim3(256), 0, (hipStream_t)stream, f, wt);
  NSC_CHECK_LAUNCH("weight_flip_transpose4");
  return NSC_OK;
}
extern "C" int nsc_weight_flip_transpose(const float* w, float* wt, int K, int Cin, int Cout, void* stream) {
  NSC_REQUIRE(w && wt && K > 0 && Cin > 0 && Cout > 0, NSC_ERR_BAD_ARG, "nsc_weight_flip_transpose: bad args");
  const int n = K * Cin * Cout;
  hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3(std::min(1024, nsc_cdiv(n, 256))), dim3(256), 0,
                     (hipStream_t)stream, w, wt, K, Cin, Cout);
  NSC_CHECK_LAUNCH("weight_flip_transpose");
  return NSC_OK;
}
