// quant.hip - soft-to-hard scalar quantizer (nn_core_operator.py:140-164 of the reference) fused with the
// quan_loss / entropy_coding_loss partials (loss_terms_and_measures.py:257-267).  HBM-bound.
//
// Mapping: one workgroup per frame; LPC lanes cooperate on one code (each lane owns 4*ITER bins kept in
// registers), so a wave handles 64/LPC codes per pass and its float4 store of p is one contiguous
// 1-KiB segment.  max / sum / dot reductions are xor-shuffles inside the LPC-lane group; the histogram and
// d(bins) are accumulated in registers across a frame, merged with LDS float atomics, then one global
// atomic per bin per workgroup.
#include "nsc_common.h"
#include "quant_common.h"
#include <type_traits>
#include <algorithm>

#ifndef NSC_QWGRID
#define NSC_QWGRID NSC_PROBE_INT("NSC_QWGRID", 512)   // workgroups (4 waves = 4 frames in flight each) of the wave-per-frame forward kernel: 512 -> 5.06, 1024 -> 4.8, 256 -> 4.2 TB/s at B = 4096
#endif
#ifndef NSC_QGRID
#define NSC_QGRID 1024   // workgroups of the forward kernel (measured at B = 4096: 512 -> 3.7, 768 -> 4.05, 1024 -> 4.08, 1280 -> 3.75, 2048 -> 3.3 TB/s)
#endif

// FULL: nb == 4*LPC*ITER (32 bins on 8 lanes, 256 on 64): every (lane, slot) holds a real bin, so the masks, selects and
// scalar-store fallbacks fold away (they made ~2/3 of the pass's VALU instructions)
template <int LPC, int ITER, bool FULL>
__global__ __launch_bounds__(256) void quantize_fwd_kernel(const float* __restrict__ code,
                                                           const float* __restrict__ alpha_p,
                                                           const float* __restrict__ bins, float on, int soft, int L,
                                                           int nb, float* __restrict__ p_out, float* __restrict__ out,
                                                           float* __restrict__ quan_out, float* __restrict__ hist,
                                                           int nbatch) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [nbpad] histogram + [4] quan partials
  constexpr int CPW = 64 / LPC;
  const int nbpad = 4 * LPC * ITER;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane % LPC, gc = lane / LPC;
  const int B = (int)nbatch;
  const float alpha = alpha_p[0];
  float bv[ITER][4];
  bool ok[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = (i * LPC + gl) * 4 + j;
      ok[i][j] = FULL || k < nb;
      bv[i][j] = ok[i][j] ? bins[k] : 0.f;
    }
  for (int k = tid; k < nbpad + 4; k += 256) sh[k] = 0.f;
  __syncthreads();
  float hacc[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) hacc[i][j] = 0.f;
  const bool vec_ok = FULL || (nb & 3) == 0;
  // A workgroup walks frames b, b + gridDim.x, ... : the histogram is accumulated across ALL its frames before the
  // flush - one atomic per bin per workgroup with <= 1024 workgroups, instead of one per bin per FRAME (4096
  // serialised atomics on each of the 32 addresses bounded the first version at large batch).
  // The codes of a frame are fetched one frame AHEAD (NPF loads in flight per lane): with the load inside the pass loop
  // every pass of 64/LPC codes waited a full memory round trip on its own code (8 passes x ~1 us per frame: the kernel
  // ran at 2.8 TB/s with the VALU half idle).  (Staging them through LDS instead measured slower: 3.4 vs 3.8 TB/s.)
  constexpr int NPF = 8;
  float cn[NPF];
  auto fetch_codes = [&](int bf) {
    const int bc = bf < B ? bf : blockIdx.x;
#pragma unroll
    for (int it = 0; it < NPF; ++it) {
      const int l = (wave + 4 * it) * CPW + gc;
      cn[it] = code[(long)bc * L + (l < L ? l : 0)];
    }
  };
  fetch_codes(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
  float qacc = 0.f;
  float cc[NPF];
#pragma unroll
  for (int it = 0; it < NPF; ++it) cc[it] = cn[it];
  fetch_codes(b + gridDim.x);
  int it = 0;
  for (int l0 = wave * CPW; l0 < L; l0 += 4 * CPW, ++it) {
    const int l = l0 + gc;
    const bool live = l < L;
    const long ci = (long)b * L + (live ? l : 0);
    float c;
    if (it < NPF) {
      c = cc[0];
#pragma unroll
      for (int q = 1; q < NPF; ++q) c = (it == q) ? cc[q] : c;     // wave-uniform select (it is uniform)
    } else {
      c = code[ci];
    }
    float dist[ITER][4], p[ITER][4];
    softmax_bins<LPC, ITER>(c, alpha, bv, ok, dist, p);
    float q;
    if (soft) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s = fmaf(p[i][j], bv[i][j], s);
      q = grp_sum<LPC>(s);
    } else {
      float best = -1.f;
      int idx = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = (i * LPC + gl) * 4 + j;
          if (ok[i][j] && p[i][j] > best) { best = p[i][j]; idx = k; }
        }
      idx = grp_argmax<LPC>(best, idx);
      // a NaN code makes every p NaN: no comparison succeeds and idx keeps its sentinel - never index the table with it, and a
      // diverged run stays visibly NaN (like nsc_tanh)
      q = (c != c) ? c : bins[min(idx, nb - 1)];
    }
    if (live) {
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hacc[i][j] += p[i][j];
          if (ok[i][j]) qacc += __builtin_amdgcn_sqrtf(p[i][j] + QEPS);   // v_sqrt_f32 (1 ulp); p + 1e-20 is never denormal
        }
      if (p_out) {
        float* pr = p_out + ci * nb;
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
          const int k0 = (i * LPC + gl) * 4;
          if (vec_ok) {
            if (k0 < nb) *reinterpret_cast<float4*>(pr + k0) = make_float4(p[i][0], p[i][1], p[i][2], p[i][3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (ok[i][j]) pr[k0 + j] = p[i][j];
          }
        }
      }
      if (gl == 0) out[ci] = (1.f - on) * c + on * q;
    }
  }
  // per-frame quan_loss partial
  qacc = wave_sum(qacc);
  if (lane == 0) sh[nbpad + wave] = qacc;
  __syncthreads();
  if (quan_out && tid == 0)
    quan_out[b] = (sh[nbpad] + sh[nbpad + 1] + sh[nbpad + 2] + sh[nbpad + 3]) / (float)L;
  __syncthreads();
  }  // frames
  // merge the histogram of all frames of this workgroup
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = (i * LPC + gl) * 4 + j;
      if (ok[i][j]) atomicAdd(&sh[k], hacc[i][j]);
    }
  __syncthreads();
  if (hist)
    for (int k = tid; k < nb; k += 256) atomicAdd(hist + k, sh[k]);
}

template <int LPC, int ITER>
__global__ __launch_bounds__(256) void quantize_bwd_kernel(
    const float* __restrict__ code, const float* __restrict__ alpha_p, const float* __restrict__ bins, float on,
    int soft, int L, int nb, const float* __restrict__ dout, const float* __restrict__ dp, float cq,
    const float* __restrict__ ghist, float ent_scale, int pre_tanh, float* __restrict__ dcode,
    float* __restrict__ dalpha, float* __restrict__ dbins) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [nbpad] dbins + [4] dalpha partials
  constexpr int CPW = 64 / LPC;
  const int nbpad = 4 * LPC * ITER;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane % LPC, gc = lane / LPC;
  const int b = blockIdx.x;
  const float alpha = alpha_p[0];
  float bv[ITER][4], gh[ITER][4];
  bool ok[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = (i * LPC + gl) * 4 + j;
      ok[i][j] = k < nb;
      bv[i][j] = ok[i][j] ? bins[k] : 0.f;
      gh[i][j] = (ok[i][j] && ghist) ? ent_scale * ghist[k] : 0.f;
    }
  for (int k = tid; k < nbpad + 4; k += 256) sh[k] = 0.f;
  __syncthreads();
  float dbacc[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) dbacc[i][j] = 0.f;
  float daacc = 0.f;
  // the codes / upstream gradients of the first NPF passes are fetched up front (NPF x 2 loads in flight per lane): with the
  // loads inside the pass loop every pass waited a memory round trip of its own (8 passes per frame at L = 256: the
  // kernel was ~14 us of latency at B = 128)
  constexpr int NPF = LPC >= 64 ? 4 : 8;     // passes per frame in the shapes used: L / (4 * 64 / LPC) = 8 (32 bins, L 256) | 4 (LSF)
  float cpre[NPF], gpre[NPF];
#pragma unroll
  for (int it = 0; it < NPF; ++it) {
    const int l = (wave + 4 * it) * CPW + gc;
    const long ci = (long)b * L + (l < L ? l : 0);
    cpre[it] = code[ci];
    gpre[it] = dout ? dout[ci] : 0.f;
  }
  int it = 0;
  for (int l0 = wave * CPW; l0 < L; l0 += 4 * CPW, ++it) {
    const int l = l0 + gc;
    const bool live = l < L;
    const long ci = (long)b * L + (live ? l : 0);
    float c, go;
    if (it < NPF) {
      c = cpre[0]; go = gpre[0];
#pragma unroll
      for (int q = 1; q < NPF; ++q) {                     // wave-uniform select (it is uniform)
        c = (it == q) ? cpre[q] : c;
        go = (it == q) ? gpre[q] : go;
      }
    } else {
      c = code[ci];
      go = dout ? dout[ci] : 0.f;
    }
    go = live ? go : 0.f;
    float dist[ITER][4], p[ITER][4], gp[ITER][4];
    softmax_bins<LPC, ITER>(c, alpha, bv, ok, dist, p);
    int hidx = -1;
    if (!soft) {
      float best = -1.f;
      int idx = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < ITER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = (i * LPC + gl) * 4 + j;
          if (ok[i][j] && p[i][j] > best) { best = p[i][j]; idx = k; }
        }
      hidx = grp_argmax<LPC>(best, idx);
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float g = 0.f;
        if (ok[i][j] && live) {
          g = gh[i][j] + cq * 0.5f * __builtin_amdgcn_rsqf(p[i][j] + QEPS);   // v_rsq_f32 (1 ulp)
          if (dp) g += dp[ci * nb + (i * LPC + gl) * 4 + j];
          if (soft) g = fmaf(on * go, bv[i][j], g);
        }
        gp[i][j] = g;
        dot = fmaf(p[i][j], g, dot);
      }
    dot = grp_sum<LPC>(dot);
    float dcp = 0.f;
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gz = p[i][j] * (gp[i][j] - dot);
        daacc = fmaf(gz, dist[i][j], daacc);
        const float df = c - bv[i][j];
        const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        const float gd = gz * alpha * sgn;
        dcp += gd;
        float dbk = -gd;
        if (soft) dbk = fmaf(on * go, p[i][j], dbk);
        else if ((i * LPC + gl) * 4 + j == hidx) dbk += on * go;
        if (ok[i][j] && live) dbacc[i][j] += dbk;
      }
    dcp = grp_sum<LPC>(dcp);
    if (live && gl == 0 && dcode) {
      float dc = (1.f - on) * go + dcp;
      if (pre_tanh) dc *= (1.f - c * c);
      dcode[ci] = dc;
    }
  }
  if (dbins) {
#pragma unroll
    for (int i = 0; i < ITER; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = (i * LPC + gl) * 4 + j;
        if (ok[i][j]) atomicAdd(&sh[k], dbacc[i][j]);
      }
  }
  daacc = wave_sum(daacc);
  if (lane == 0) sh[nbpad + wave] = daacc;
  __syncthreads();
  if (dbins)
    for (int k = tid; k < nb; k += 256) atomicAdd(dbins + k, sh[k]);
  if (dalpha && tid == 0) atomicAdd(dalpha, sh[nbpad] + sh[nbpad + 1] + sh[nbpad + 2] + sh[nbpad + 3]);
}

// ---- the 32-bin forward with p materialised at LARGE batch (the op-surface form; config 5: 4096 frames): one WAVE per frame.
// The workgroup-per-frame kernel above writes 32 KB of p per frame and then meets at two workgroup barriers (the per-frame
// quan_loss), loads its codes as eight 32-byte pieces per lane and stores the quantised codes as eight 32-byte pieces per
// wave: 24 memory instructions per wave for 8 KB of p.  Here a wave owns a frame: it loads 64 codes with ONE coalesced
// instruction, hands them to the 8-lane groups through the cross-lane network (ds_bpermute), collects the 64 quantised
// codes the same way and stores them with ONE instruction, reduces quan_loss inside the wave - no barrier in the frame
// loop - and the 1-KiB float4 stores of p are all that is left per pass.
// LPC lanes share a code (ITER = 8 / LPC float4 groups of bins per lane).  LPC = 8 ships (p bit-equal to the workgroup kernel).
// LPC = 4 (probes library, NSC_QLPC=4: 16 codes per pass, two DPP steps per reduction, ~320 instead of ~560 lane-instructions
// per code) measured NO faster (4.6-5.1 vs 4.8-5.1 TB/s), as did single 16-byte stores instead of the 12 + 4-byte splits the
// allocator produced in half of the passes: the kernel is bound neither by its VALU count nor by the store shape; a
// write-only probe in the same grid and geometry reaches 6.8 TB/s on this (Infinity-Cache-sized) buffer.
template <bool SOFT, int LPC>
__global__ __launch_bounds__(256) void quantize_fwd32_wave_kernel(const float* __restrict__ code, const float* __restrict__ alpha_p,
                                                                  const float* __restrict__ bins, float on, int L,
                                                                  float* __restrict__ p_out, float* __restrict__ out,
                                                                  float* __restrict__ quan_out, float* __restrict__ hist, int B,
                                                                  int nostore) {
  // nostore (probes build only, NSC_QUANT_NO_STORE=1; 0 in the shipped library): the p stores are issued against an EMPTY
  // buffer descriptor and dropped by the bounds check - the kernel's arithmetic without its 1-KiB stores reaching memory
  constexpr int NB = 32, ITER = NB / (4 * LPC), CPW = 64 / LPC, NPASS = 64 / CPW;
  __shared__ float sh[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gl = lane % LPC, gc = lane / LPC;
  const float alpha = alpha_p[0];
  float bv[ITER][4];
  bool ok[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bv[i][j] = bins[(i * LPC + gl) * 4 + j];
      ok[i][j] = true;
    }
  if (tid < NB) sh[tid] = 0.f;
  __syncthreads();
  float hacc[ITER][4];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) hacc[i][j] = 0.f;
  const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
  // fast path of the pass (alpha <= 0, see below): ah = |alpha| log2(e) / 2 and this lane's bins as float pairs
  const bool fast = alpha <= 0.f;                           // wave-uniform (NaN alpha: the generic path keeps it visible)
  const float ahalf = -0.5f * 1.4426950408889634f * alpha;
  f32x2 bp[ITER][2];
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) bp[i][h] = (f32x2){bv[i][2 * h], bv[i][2 * h + 1]};
  const int nch = L >> 6;                                   // chunks of 64 codes per frame (L is a multiple of 64 here)
  // lane X of the wave keeps the quantised code of chunk element X: at pass X / CPW it picks it from a lane of group X % CPW
  const int pick_idx = ((lane % CPW) * LPC) * 4, pick_pass = lane / CPW;
  const __amdgpu_buffer_rsrc_t sp = __builtin_amdgcn_make_buffer_rsrc(p_out, 0, nostore ? 0u : (unsigned)((long)B * L * NB * 4), 0x00020000);
  float vn = w0 < B ? code[(long)w0 * L + lane] : 0.f;      // first chunk of the first frame
  // (the frame loop is instantiated twice and the wave-uniform choice made once, outside it)
  auto frames = [&](auto fast_c) {
    constexpr bool FAST = decltype(fast_c)::value;
    for (int f = w0; f < B; f += nwaves) {
      float qacc = 0.f;
      for (int ch = 0; ch < nch; ++ch) {
        const float v = vn;
        {   // next chunk (of this frame or of the wave's next frame): in flight during the passes below
          const int chn = ch + 1 < nch ? ch + 1 : 0;
          const long fn = ch + 1 < nch ? f : (f + nwaves < B ? f + nwaves : f);
          vn = code[fn * L + chn * 64 + lane];
        }
        float outv = 0.f;
        const long cbase = (long)f * L + ch * 64;
  #pragma unroll
        for (int it = 0; it < NPASS; ++it) {
          // code of this pass's group gc: element it * CPW + gc of the chunk
          const float c = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((it * CPW + gc) * 4, __builtin_bit_cast(int, v)));
          float p[ITER][4];
          float qsum;                                   // this lane's share of sum_k sqrt(p_k + eps)
          if constexpr (FAST) {
            // ROUND 4.  With its p stores dropped this kernel still took 25.5 of its 29 us (tools/quant_time.py, NSC_QUANT_NO_STORE):
            // it was bound by VALU issue - 17 instruction slots per bin, two of them quarter-rate transcendentals (v_exp for p,
            // v_sqrt for the quan_loss term) - not by HBM.  For alpha <= 0 (the reference's regime: init -300, and the only one
            // in which the soft assignment approaches a hard one) the pass is restated so that ONE transcendental per bin is left
            // and most of the rest runs on packed fp32 (v_pk_*: two floats per lane and instruction):
            //   x_k = ah (c - b_k)  (ah = |alpha| log2(e) / 2 >= 0, so |x_k| = ah |c - b_k| bit for bit: codes at EQUAL distance
            //   from two bins still tie exactly and the hard assignment keeps tf.nn.top_k's lowest index)   M = min_k |x_k|
            //   t_k = 2^(M - |x_k|)  = sqrt of the softmax numerator;  S = sum_k t_k^2
            //   p_k = t_k^2 / S      sqrt(p_k) = t_k rsqrt(S)          (sqrt(p + 1e-20) - sqrt(p) <= 1e-10: dropped)
            // abs / neg ride on source modifiers.  p differs from the exp / rcp form by rounding only (~3 ulp).
            const f32x2 c2 = {c, c}, ah2 = {ahalf, ahalf};
            f32x2 x[ITER][2];
            float mn = INFINITY;
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int h = 0; h < 2; ++h) {
                x[i][h] = (c2 - bp[i][h]) * ah2;
                mn = fminf(mn, fminf(fabsf(x[i][h][0]), fabsf(x[i][h][1])));
              }
            mn = grp_min8<LPC>(mn);
            f32x2 t[ITER][2], s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int h = 0; h < 2; ++h) {
                t[i][h] = (f32x2){__builtin_amdgcn_exp2f(mn - fabsf(x[i][h][0])), __builtin_amdgcn_exp2f(mn - fabsf(x[i][h][1]))};
                s1 += t[i][h];
                s2 = __builtin_elementwise_fma(t[i][h], t[i][h], s2);
              }
            const float S = grp_sum<LPC>(s2[0] + s2[1]);
            const float inv = __builtin_amdgcn_rcpf(S);
            qsum = (s1[0] + s1[1]) * __builtin_amdgcn_rsqf(S);
            const f32x2 inv2 = {inv, inv};
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int h = 0; h < 2; ++h) {
                const f32x2 pp = (t[i][h] * inv2) * t[i][h];
                p[i][2 * h] = pp[0];
                p[i][2 * h + 1] = pp[1];
              }
          } else {
            float dist[ITER][4];
            softmax_bins<LPC, ITER>(c, alpha, bv, ok, dist, p);
            qsum = 0.f;
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int j = 0; j < 4; ++j) qsum += __builtin_amdgcn_sqrtf(p[i][j] + QEPS);
          }
          qacc += qsum;
          float q;
          if (SOFT) {
            float s_ = 0.f;
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int j = 0; j < 4; ++j) s_ = fmaf(p[i][j], bv[i][j], s_);
            q = grp_sum<LPC>(s_);
          } else {
            float best = -1.f;
            int idx = 0x7fffffff;
  #pragma unroll
            for (int i = 0; i < ITER; ++i)
  #pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int k = (i * LPC + gl) * 4 + j;
                if (p[i][j] > best) { best = p[i][j]; idx = k; }
              }
            idx = grp_argmax<LPC>(best, idx);
            q = (c != c) ? c : bins[min(idx, NB - 1)];     // NaN in: NaN out, and the sentinel never indexes the table
          }
  #pragma unroll
          for (int i = 0; i < ITER; ++i) {
  #pragma unroll
            for (int j = 0; j < 4; ++j) hacc[i][j] += p[i][j];
            // ONE 16-byte store per lane and float4 group (a plain float4 assignment was split into a 12-byte and a 4-byte store
            // in half of the unrolled passes, where the allocator had not kept the four values in consecutive registers)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, (f32x4){p[i][0], p[i][1], p[i][2], p[i][3]}), sp,
                                                   (int)((cbase + it * CPW + gc) * (NB * 4) + (i * LPC + gl) * 16), 0, 0);
          }
          // the group's quantised code travels to lane it * CPW + gc
          const float o = (1.f - on) * c + on * q;
          const float t = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(pick_idx, __builtin_bit_cast(int, o)));
          outv = pick_pass == it ? t : outv;
        }
        out[cbase + lane] = outv;
      }
      qacc = wave_sum(qacc);
      if (quan_out && lane == 0) quan_out[f] = qacc / (float)L;
    }
  };
  if (fast) frames(std::true_type{});
  else frames(std::false_type{});
#pragma unroll
  for (int i = 0; i < ITER; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&sh[(i * LPC + gl) * 4 + j], hacc[i][j]);
  __syncthreads();
  if (hist && tid < NB) atomicAdd(hist + tid, sh[tid]);
}

#ifdef NSC_PROBES
// write-only probe in the SAME grid / geometry as quantize_fwd32_wave_kernel (probes library, NSC_QUANT_WRITE_ONLY=1): every
// wave writes its frames' 32 KB of p as 1-KiB float4 wave-stores plus the 256-B rows of out, nothing else.  NT: nontemporal.
template <bool NT>
__global__ __launch_bounds__(256) void quantize_write_probe_kernel(int L, float* __restrict__ p_out, float* __restrict__ out, int B) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane & 7, gc = lane >> 3;
  const int nwaves = gridDim.x * 4, w0 = blockIdx.x * 4 + wave;
  const f32x4 v = {0.03125f, 0.03125f, 0.03125f, 0.03125f};
  for (int f = w0; f < B; f += nwaves)
    for (int ch = 0; ch < (L >> 6); ++ch) {
      const long cbase = (long)f * L + ch * 64;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        f32x4* dst = reinterpret_cast<f32x4*>(p_out + (cbase + it * 8 + gc) * 32 + gl * 4);
        if (NT) __builtin_nontemporal_store(v, dst);
        else *dst = v;
      }
      out[cbase + lane] = 0.5f;
    }
}
#endif

// pick (LPC, ITER) for nb bins
#define QDISPATCH(NB, CALL)                                    \
  do {                                                         \
    const int q4__ = ((NB) + 3) / 4;                           \
    if (q4__ <= 1) { CALL(1, 1); }                             \
    else if (q4__ <= 2) { CALL(2, 1); }                        \
    else if (q4__ <= 4) { CALL(4, 1); }                        \
    else if (q4__ <= 8) { CALL(8, 1); }                        \
    else if (q4__ <= 16) { CALL(16, 1); }                      \
    else if (q4__ <= 32) { CALL(32, 1); }                      \
    else if (q4__ <= 64) { CALL(64, 1); }                      \
    else if (q4__ <= 128) { CALL(64, 2); }                     \
    else { CALL(64, 4); }                                      \
  } while (0)

extern "C" int nsc_quantize_fwd(const float* code, const float* alpha, const float* bins, float is_quan_on, int soft,
                                int B, int L, int nb, float* p_out, float* out, float* quan_out, float* hist,
                                void* stream) {
  NSC_REQUIRE(code && alpha && bins && out, NSC_ERR_BAD_ARG, "nsc_quantize_fwd: null code/alpha/bins/out");
  NSC_REQUIRE(B > 0 && L > 0 && nb > 0, NSC_ERR_BAD_ARG, "nsc_quantize_fwd: non-positive B/L/nb");
  NSC_REQUIRE(nb <= 1024, NSC_ERR_UNSUPPORTED, "nsc_quantize_fwd: nb %d > 1024", nb);
  hipStream_t st = (hipStream_t)stream;
  if (nb == 32 && p_out && (L & 63) == 0 && B >= 1024 && (long)B * L * 128 < (1L << 31) && !NSC_PROBE_SET("NSC_QUANT_WG")) {
    // op-surface form at inference batch sizes: one wave per frame (see quantize_fwd32_wave_kernel)
    const int grid = std::min((B + 3) / 4, NSC_QWGRID);
#ifdef NSC_PROBES
    if (NSC_PROBE_SET("NSC_QUANT_WRITE_ONLY")) {
      if (NSC_PROBE_INT("NSC_QUANT_WRITE_ONLY", 1) == 2) hipLaunchKernelGGL(quantize_write_probe_kernel<true>, dim3(grid), dim3(256), 0, st, L, p_out, out, B);
      else hipLaunchKernelGGL(quantize_write_probe_kernel<false>, dim3(grid), dim3(256), 0, st, L, p_out, out, B);
      return NSC_OK;
    }
#endif
#define QWAVE(SOFT_, LPC_) hipLaunchKernelGGL((quantize_fwd32_wave_kernel<SOFT_, LPC_>), dim3(grid), dim3(256), 0, st, code, alpha, bins, \
                                              is_quan_on, L, p_out, out, quan_out, hist, B, NSC_PROBE_INT("NSC_QUANT_NO_STORE", 0))
    if (NSC_PROBE_INT("NSC_QLPC", 8) == 8) { if (soft) QWAVE(true, 8); else QWAVE(false, 8); }
    else { if (soft) QWAVE(true, 4); else QWAVE(false, 4); }
#undef QWAVE
    NSC_CHECK_LAUNCH("quantize_fwd (wave per frame)");
    return NSC_OK;
  }
#define CALLF(LPC_, IT_)                                                                                          \
  do {                                                                                                            \
    if (nb == 4 * LPC_ * IT_)                                                                                     \
      hipLaunchKernelGGL((quantize_fwd_kernel<LPC_, IT_, true>), dim3(std::min(B, NSC_QGRID)), dim3(256),          \
                         (4 * LPC_ * IT_ + 4) * sizeof(float), st, code, alpha, bins, is_quan_on, soft, L, nb, p_out,  \
                         out, quan_out, hist, B);                                                                  \
    else                                                                                                          \
      hipLaunchKernelGGL((quantize_fwd_kernel<LPC_, IT_, false>), dim3(std::min(B, NSC_QGRID)), dim3(256),         \
                         (4 * LPC_ * IT_ + 4) * sizeof(float), st, code, alpha, bins, is_quan_on, soft, L, nb, p_out,  \
                         out, quan_out, hist, B);                                                                  \
  } while (0)
  QDISPATCH(nb, CALLF);
#undef CALLF
  NSC_CHECK_LAUNCH("quantize_fwd");
  return NSC_OK;
}

extern "C" int nsc_quantize_bwd(const float* code, const float* alpha, const float* bins, float is_quan_on, int soft,
                                int B, int L, int nb, const float* dout, const float* dp, float c_quan,
                                const float* ghist, float ent_scale, int pre_tanh, float* dcode, float* dalpha,
                                float* dbins, void* stream) {
  NSC_REQUIRE(code && alpha && bins, NSC_ERR_BAD_ARG, "nsc_quantize_bwd: null code/alpha/bins");
  NSC_REQUIRE(B > 0 && L > 0 && nb > 0, NSC_ERR_BAD_ARG, "nsc_quantize_bwd: non-positive B/L/nb");
  NSC_REQUIRE(nb <= 1024, NSC_ERR_UNSUPPORTED, "nsc_quantize_bwd: nb %d > 1024", nb);
  hipStream_t st = (hipStream_t)stream;
  const float cq = c_quan / (float)L;
#define CALLB(LPC_, IT_)                                                                                          \
  hipLaunchKernelGGL((quantize_bwd_kernel<LPC_, IT_>), dim3(B), dim3(256), (4 * LPC_ * IT_ + 4) * sizeof(float), st, \
                     code, alpha, bins, is_quan_on, soft, L, nb, dout, dp, cq, ghist, ent_scale, pre_tanh, dcode,   \
                     dalpha, dbins)
  QDISPATCH(nb, CALLB);
#undef CALLB
  NSC_CHECK_LAUNCH("quantize_bwd");
  return NSC_OK;
}

// entropy_coding_loss from the batch histogram (loss_terms_and_measures.py:262-267) + its gradient wrt hist
__device__ __forceinline__ void entropy_from_hist_body(const float* __restrict__ hist, int nb, float* __restrict__ ent,
                                                       float* __restrict__ ghist);
__global__ __launch_bounds__(256) void entropy_from_hist_kernel(const float* __restrict__ hist, int nb,
                                                                float* __restrict__ ent, float* __restrict__ ghist) {
  entropy_from_hist_body(hist, nb, ent, ghist);
}
// several quantizers' histograms in one launch (one workgroup each): a step has one per codec + the LSF quantizer's
struct EntropyBatch {
  nsc_entropy_job j[NSC_ENT_MAXJ];
};
__global__ __launch_bounds__(256) void entropy_from_hist_batch_kernel(EntropyBatch t) {
  nsc_entropy_job jb = t.j[0];
#pragma unroll
  for (int q = 1; q < NSC_ENT_MAXJ; ++q)
    if (q == (int)blockIdx.x) jb = t.j[q];
  entropy_from_hist_body(jb.hist, jb.nb, jb.ent, jb.ghist);
}
__device__ __forceinline__ void entropy_from_hist_body(const float* __restrict__ hist, int nb, float* __restrict__ ent,
                                                       float* __restrict__ ghist) {
  __shared__ float red[4];
  __shared__ float bc[2];
  const int tid = threadIdx.x;
  const float kInvLn2 = 1.4426950408889634f;
  float s = 0.f;
  for (int k = tid; k < nb; k += 256) s += hist[k];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) bc[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float S = bc[0];
  float e = 0.f, gh = 0.f;
  for (int k = tid; k < nb; k += 256) {
    const float h = hist[k] / S;
    const float lg = logf(h + 1e-7f);
    e -= h * lg * kInvLn2;
    const float g = -(lg + h / (h + 1e-7f)) * kInvLn2;
    gh += g * h;
  }
  e = wave_sum(e);
  gh = wave_sum(gh);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = e;
  __syncthreads();
  if (tid == 0 && ent) ent[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = gh;
  __syncthreads();
  if (tid == 0) bc[1] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if (ghist) {
    const float gdot = bc[1];
    for (int k = tid; k < nb; k += 256) {
      const float h = hist[k] / S;
      const float g = -(logf(h + 1e-7f) + h / (h + 1e-7f)) * kInvLn2;
      ghist[k] = (g - gdot) / S;
    }
  }
}
// Per-FRAME entropy of the soft assignments: what the reference's validation loop reads when it evaluates
// entropy_coding_loss one frame at a time (neural_speech_coding_module.py:685-722: batch of 1 => the histogram is that of
// a single frame).  ent[b] = -sum_k h log2(h + 1e-7), h = sum_l p[b,l,k] / sum_{l,k} p[b,l,k].  One workgroup per frame.
__global__ __launch_bounds__(256) void frame_entropy_kernel(const float* __restrict__ p, int L, int nb,
                                                            float* __restrict__ ent) {
  extern __shared__ float sh[];           // [nb] histogram + [256] reduction scratch
  float* red = sh + nb;
  const int tid = threadIdx.x;
  const float* pb = p + (long)blockIdx.x * L * nb;
  for (int k = tid; k < nb; k += 256) sh[k] = 0.f;
  __syncthreads();
  // thread t owns bin (t % nb) of rows t / nb, t / nb + 256 / nb, ... when nb divides 256; otherwise a plain strided walk
  const int n = L * nb;
  if (256 % nb == 0) {
    const int k = tid % nb, rows = 256 / nb;
    float s = 0.f;
    for (int l = tid / nb; l < L; l += rows) s += pb[l * nb + k];
    atomicAdd(&sh[k], s);
  } else {
    for (int e = tid; e < n; e += 256) atomicAdd(&sh[e % nb], pb[e]);
  }
  __syncthreads();
  float s = 0.f;
  for (int k = tid; k < nb; k += 256) s += sh[k];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const float S = red[0];
  __syncthreads();
  float a = 0.f;
  for (int k = tid; k < nb; k += 256) {
    const float h = sh[k] / S;
    a -= h * logf(h + 1e-7f);
  }
  red[tid] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) ent[blockIdx.x] = red[0] * 1.4426950408889634f;
}
extern "C" int nsc_frame_entropy(const float* p, int B, int L, int nb, float* ent, void* stream) {
  NSC_REQUIRE(p && ent && B > 0 && L > 0 && nb > 0, NSC_ERR_BAD_ARG, "nsc_frame_entropy: bad args");
  hipLaunchKernelGGL(frame_entropy_kernel, dim3(B), dim3(256), (nb + 256) * sizeof(float), (hipStream_t)stream, p, L, nb, ent);
  NSC_CHECK_LAUNCH("frame_entropy");
  return NSC_OK;
}

extern "C" int nsc_entropy_from_hist_batch(const nsc_entropy_job* jobs, int njobs, void* stream) {
  NSC_REQUIRE(jobs && njobs > 0 && njobs <= NSC_ENT_MAXJ, NSC_ERR_BAD_ARG, "nsc_entropy_from_hist_batch: 1..%d jobs", NSC_ENT_MAXJ);
  EntropyBatch t;
  for (int q = 0; q < NSC_ENT_MAXJ; ++q) {
    t.j[q] = jobs[q < njobs ? q : 0];
    NSC_REQUIRE(t.j[q].hist && t.j[q].nb > 0, NSC_ERR_BAD_ARG, "nsc_entropy_from_hist_batch: job %d: bad args", q);
  }
  hipLaunchKernelGGL(entropy_from_hist_batch_kernel, dim3(njobs), dim3(256), 0, (hipStream_t)stream, t);
  NSC_CHECK_LAUNCH("entropy_from_hist_batch");
  return NSC_OK;
}

extern "C" int nsc_entropy_from_hist(const float* hist, int nb, float* ent, float* ghist, void* stream) {
  NSC_REQUIRE(hist && nb > 0, NSC_ERR_BAD_ARG, "nsc_entropy_from_hist: bad args");
  hipLaunchKernelGGL(entropy_from_hist_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hist, nb, ent, ghist);
  NSC_CHECK_LAUNCH("entropy_from_hist");
  return NSC_OK;
}
