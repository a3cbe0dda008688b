// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC fp32 activations,
// [Cout][kh][kw][Cin] weights, exact-f32 MFMA (v_mfma_f32_32x32x2_f32),
// fused bias(+folded BN) + residual / nearest-2x top-down add + ReLU epilogue.
//
// Replaces tf.nn.conv2d + tf.nn.batch_normalization + relu/add on the path
// (reference nn.py:337-381, :459-521, :947-1014, :1771-1774) and tf.matmul of
// the box head (nn.py:730-774; an FC is a 1x1 conv over "pixels" = RoIs).
//
// GEMM view: C[m][n] = sum_k A[m][k] * W[n][k],  m = (img,ho,wo), n = cout,
// k = (kh,kw,ci).  Both operands are K-contiguous, so a BK=32 slice of either
// is one 128-byte row read (8 lanes x 16 B), gathered on the fly for A.
//
// Tiling (wave64): 256 threads = 4 waves as WM x WN, each wave owns TM x TN
// MFMA tiles of 32x32 -> block tile BM x BN = (WM*TM*32) x (WN*TN*32), BK = 32.
// LDS: two stages of [BM+BN][36] floats (row stride 144 B = 128 + one 16-B
// access: conflict-free ds_read_b128 over the 16-lane groups and conflict-free
// ds_write_b128).  K order inside a 32-slice is permuted (k = 8g + 4*(lane>>5)
// + t) for BOTH operands so one ds_read_b128 feeds four MFMA k-steps.
// Pipeline: global->register prefetch of slice c+1 is issued before the 16*TM*TN
// MFMAs on slice c, written to the other LDS stage after them; one barrier
// per slice.
#include <cstdlib>
#include <type_traits>

#include "odt_common.hpp"

namespace odt {

namespace {

constexpr int LS = 36;  // LDS row stride in floats (32 + 4 pad)

typedef unsigned int u32x4 __attribute__((vector_size(16)));

// exact n / d for n < 2^31 with the (mul, sh) pair made by conv_prepare (mul == 0: d == 1)
__device__ __forceinline__ int fast_div(int n, unsigned mul, unsigned sh) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}
constexpr unsigned kOOB = 0x80000000u;   // buffer offset that is out of range for every tensor (< 2 GiB)

// ST = number of LDS stages.  ST == 2: double-buffered stages + rotated one-barrier pipeline
// (2 workgroups per CU).  ST == 1: one stage (36.9 KB for the 128x128 tile, registers prefetch
// one slice ahead, two barriers per slice) so that three workgroups fit on a CU and another
// workgroup's MFMA stream covers this one's barriers, prologue and HBM-bound epilogue.
template <int WM, int WN, int TM, int TN, int ST>
__global__ void __launch_bounds__(256, ST == 1 ? 3 : 2) conv_igemm_kernel(const ConvParams* __restrict__ pp) {
  // Parameters live in device memory (one record per conv of the plan): the by-value kernarg
  // block sits in host-coherent memory and its cold scalar loads cost the first dispatch wave of
  // every launch tens of microseconds.
  const ConvParams p = *pp;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float lds[ST][(BM + BN) * LS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.Cout + BN - 1) / BN;
  auto stamp = [&](int i) {
    if (p.trace != nullptr && tid == 0) p.trace[(size_t)blockIdx.x * 16 + i] = wall_clock64();
  };
  stamp(0);
  if (p.trace != nullptr && tid == 0) {       // placement: HW_ID (cu/sh/se/tg slot) and XCC_ID
    p.trace[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    p.trace[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
  // XCD-aware tile order: the dispatcher places workgroup i on XCD i % 8, each XCD has its own
  // L2.  Give every XCD one contiguous run of (m-tile, n-tile) pairs so that all N-tiles of an
  // M-tile (same A rows) and neighbouring M-tiles (shared 3x3 halo rows) hit the same L2
  // instead of pulling the A tile through the fabric once per XCD.  Bijective for any grid size;
  // a different placement would only change speed.
  int wg = (int)blockIdx.x;
  if (!(p.debug & 128)) {
    const int nwg = (int)gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int K = p.kh * p.kw * p.Cin;
  const int cpt = p.Cin >> 5;               // 32-channel slices per tap
  const int nslices = p.kh * p.kw * cpt;

  // Branch-free operand fetch: raw buffer loads (SRSRC descriptors built from kernel arguments,
  // i.e. provably wave-uniform); rows that fall into the zero padding / past M / past Cout get an
  // out-of-range offset and the hardware returns zeros.
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt, 0, (int)((unsigned)p.Cout * K * 4u), 0x00020000);

  // ---- loader role: thread -> (row lr + 32*j, 16-byte column lc)
  const int lc = tid & 7, lr = tid >> 3;
  int a_hi0[RA], a_wi0[RA];
  unsigned a_img[RA];          // byte offset of the image of this row (kOOB: row >= M)
  // 1x1 / stride 1 / unpadded convs over a dense input (all bottleneck 1x1s, FPN laterals, FCs):
  // input pixel == output row m, no (n,ho,wo) decomposition (saves the integer divisions)
  const bool dense_in = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                        p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + lr + 32 * j;
    const bool ok = m < M;
    if (dense_in) {
      a_hi0[j] = 0;
      a_wi0[j] = 0;
      a_img[j] = ok ? (unsigned)m * p.in_ldc * 4u + lc * 16u : kOOB;
    } else {
      const int mm = ok ? m : 0;
      const int n = fast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = fast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      a_hi0[j] = ho * p.stride - p.pad_t;
      a_wi0[j] = wo * p.stride - p.pad_l;
      a_img[j] = ok ? (unsigned)n * p.in_Ha * p.in_Wa * p.in_ldc * 4u + lc * 16u : kOOB;
    }
  }
  unsigned b_off[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + lr + 32 * j;
    b_off[j] = n < p.Cout ? (unsigned)n * K * 4u + lc * 16u : kOOB;
  }
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;

  const int dbg = p.debug;
  // load-stream state (runs up to two slices ahead of the MFMA stream)
  int l_cc = 0, l_kh = 0, l_kw = 0;
  unsigned l_k = 0;             // byte offset of the slice inside a weight row
  unsigned a_row[RA];           // byte offset of this tap's pixel for each row (kOOB if padded)
  auto set_tap = [&](int khh, int kww) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int hi = a_hi0[j] + khh * p.dil, wi = a_wi0[j] + kww * p.dil;
      const bool v = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && a_img[j] != kOOB;
      a_row[j] = v ? a_img[j] + (unsigned)(hi * p.in_Wa + wi) * pix_bytes : kOOB;
    }
  };
  set_tap(0, 0);

  f32x4 ra[RA], rb[RB];
  auto load_slice = [&]() {     // fetch the next slice of the stream into ra / rb
#pragma unroll
    for (int j = 0; j < RA; ++j)
      ra[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(a_row[j] + (unsigned)l_cc * 128u), 0, 0);
#pragma unroll
    for (int j = 0; j < RB; ++j)
      rb[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_wt, (int)(b_off[j] + l_k), 0, 0);
    l_k += 128u;
    if (++l_cc == cpt) {
      l_cc = 0;
      if (++l_kw == p.kw) { l_kw = 0; ++l_kh; }
      set_tap(l_kh, l_kw);      // harmless past the last tap (never loaded)
    }
  };
  auto store_slice = [&](int buf) {
    float* A = lds[buf];
    float* Bm = lds[buf] + BM * LS;
#pragma unroll
    for (int j = 0; j < RA; ++j) *reinterpret_cast<f32x4*>(&A[(lr + 32 * j) * LS + lc * 4]) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j) *reinterpret_cast<f32x4*>(&Bm[(lr + 32 * j) * LS + lc * 4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // MFMA fragments, double buffered over the four k-groups (g) of a slice.
  const int frag_off = (lane & 31) * LS + (lane >> 5) * 4;
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto read_frags = [&](int buf, int g, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    const float* A = lds[buf] + (wm * TM * 32) * LS + frag_off + g * 8;
    const float* Bm = lds[buf] + BM * LS + (wn * TN * 32) * LS + frag_off + g * 8;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LS);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bm + j * 32 * LS);
  };
  auto mfma_group = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], int t0, int t1) {
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
  };

  // Software pipeline.  Per slice c (LDS stage c&1) the four k-groups run as
  //   g0: read frags g1 | MFMA g0        g1: read frags g2 | MFMA g1
  //   g2: read frags g3 | MFMA g2 | write slice c+1 (registers -> other LDS stage)
  //   g3: barrier | read frags g0 of slice c+1 | fetch slice c+2 (global -> registers) | MFMA g3
  // so every LDS / global access has >= one k-group (16*TM*TN MFMAs) of latency cover and the
  // single barrier per slice sits in front of MFMAs whose operands are already in registers.
  // sched_barrier(0) fences pin this interleave (without them the scheduler sinks the fragment
  // reads to just before their first use and lumps the LDS writes / barrier / fetches together).
  stamp(6);
  load_slice();
  store_slice(0);
  stamp(7);
  __syncthreads();
  stamp(1);
  if (nslices > 1) load_slice();
  read_frags(0, 0, fa0, fb0);
  if constexpr (ST == 2) {
    for (int c = 0; c < nslices; ++c) {
      const int cur = c & 1;
      const bool more = c + 1 < nslices;
      if (!(dbg & 8)) read_frags(cur, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 8)) read_frags(cur, 2, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 8)) read_frags(cur, 3, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa0, fb0, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(dbg & 2)) store_slice(cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa0, fb0, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(dbg & 8)) read_frags(cur ^ 1, 0, fa0, fb0);
      if (c + 2 < nslices && !(dbg & 1)) load_slice();
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // single LDS stage: g0..g2 as above; then barrier (all fragment reads of this slice done),
    // registers -> LDS (slice c+1) with the g3 MFMAs covering the write, fetch slice c+2,
    // barrier, first fragment read of the next slice.
    for (int c = 0; c < nslices; ++c) {
      const bool more = c + 1 < nslices;
      if (!(dbg & 8)) read_frags(0, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 8)) read_frags(0, 2, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 8)) read_frags(0, 3, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(dbg & 2)) store_slice(0);
      __builtin_amdgcn_sched_barrier(0);
#ifdef ODT_AB_NOROT
      if (!(dbg & 16)) mfma_group(fa1, fb1, 0, 4);
      if (c + 2 < nslices && !(dbg & 1)) load_slice();
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(dbg & 8)) read_frags(0, 0, fa0, fb0);
#else
      if (!(dbg & 16)) mfma_group(fa1, fb1, 0, 2);
      if (c + 2 < nslices && !(dbg & 1)) load_slice();
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && !(dbg & 8)) read_frags(0, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (!(dbg & 16)) mfma_group(fa1, fb1, 2, 4);    // covers the first fragment read of slice c+1
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  stamp(2);

  // ---- epilogue.  D reg r of lane l is C[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31]: stage the
  // block tile through LDS (the A/B stages are dead) so that HBM sees whole 16-byte-per-lane
  // row segments: residual tile prefetched with independent 16-B loads, then bias + residual +
  // ReLU and 16-B stores (a wave covers full 512-byte output rows).
  constexpr int CS = BN + 4;                      // C-tile row stride (floats); +4 keeps b128 reads aligned
  // the single-stage variants stage the C tile in PASSES row blocks of RP rows (LDS is smaller)
  constexpr int PASSES = (ST == 1 && BM > 64) ? BM / 64 : 1;
  constexpr int RP = BM / PASSES;
  static_assert(RP * CS <= ST * (BM + BN) * LS, "C tile pass must fit in the A/B stages");
  static_assert(RP % (TM * 32) == 0, "a wave's rows must fall into one pass");
  float* Ct = &lds[0][0];
  constexpr int C4 = BN / 4;                      // 16-byte chunks per tile row
  constexpr int NCH = RP * C4 / 256;              // chunks per thread and pass
  static_assert(256 % C4 == 0, "bias column must be chunk-invariant");
  const bool vec_ok = (p.out_ldc & 3) == 0 && (p.res_mode == 0 || (p.res_ldc & 3) == 0);
  const bool dense_io = p.out_oy == 0 && p.out_ox == 0 && p.out_H == p.Ho && p.out_W == p.Wo &&
                        (p.res_mode == 0 || (p.res_mode == 1 && p.res_H == p.Ho && p.res_W == p.Wo));
  const bool fast = vec_ok && (p.Cout & 3) == 0;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.out, 0, (int)((unsigned)p.B * p.out_H * p.out_W * p.out_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.res_mode != 0 ? p.res : p.bias), 0,
      (int)(p.res_mode != 0 ? (unsigned)p.B * p.res_H * p.res_W * p.res_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)((unsigned)p.Cout * 4u), 0x00020000);
  const int c4 = tid % C4, row0 = tid / C4;          // chunk s2 of this thread: row0 + s2*(256/C4)
  const int col = n0 + c4 * 4;
  const bool col_ok = col < p.Cout;
  // every chunk of a thread has the same column (256 % C4 == 0): fetch the bias once, BEFORE any
  // store (a load issued after a store waits for that store's completion: vmcnt is in-order).
  f32x4 bias4 = zero4;
  if (fast) {
    bias4 = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_bias, col_ok ? col * 4 : (int)kOOB, 0, 0);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < p.Cout) bias4[e] = p.bias[col + e];
  }

  // Output / residual pixel of tile row m (dense tensors: the row index itself).
  auto out_pix = [&](int m, bool ok, unsigned& opix, unsigned& rpix) {
    if (dense_io) {
      opix = (unsigned)m;
      rpix = (unsigned)m;
    } else {
      const int mm = ok ? m : 0;
      const int n = fast_div(mm, p.div_howo_mul, p.div_howo_sh), rr = mm - n * HoWo;
      const int ho = fast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
      opix = ((unsigned)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox;
      rpix = p.res_mode == 2 ? ((unsigned)n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1)
                             : ((unsigned)n * p.res_H + ho) * p.res_W + wo;
    }
  };
  // The residual tile of ALL passes is fetched here, before the first store and before the C
  // tile is staged: vmcnt retires in order, so a load issued after a store would wait for that
  // store's write acknowledgement -- with the loads up front the passes below only read LDS and
  // fire stores, and the residual latency hides behind the staging.
  f32x4 rv[PASSES][NCH];
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass)
#pragma unroll
    for (int s2 = 0; s2 < NCH; ++s2) rv[pass][s2] = zero4;
  if (fast && p.res_mode != 0) {
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass)
#pragma unroll
      for (int s2 = 0; s2 < NCH; ++s2) {
        const int m = m0 + pass * RP + row0 + s2 * (256 / C4);
        const bool ok = col_ok && m < M;
        unsigned opix, rpix;
        out_pix(m, ok, opix, rpix);
        const unsigned roff = ok ? (rpix * p.res_ldc + col) * 4u : kOOB;
        rv[pass][s2] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)roff, 0, 0);
      }
  }

  // acc -> LDS C tile of one pass (row block of RP rows), with the barriers around it
  auto stage_pass = [&](int pass) {
    if (pass > 0) __syncthreads();                  // previous pass finished reading Ct
    if ((wm * TM * 32) / RP == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - pass * RP;
            const int cc = wn * TN * 32 + j * 32 + (lane & 31);
            Ct[row * CS + cc] = acc[i][j][r];
          }
    }
    __syncthreads();
  };

  if (fast) {
    // ---- fast path (every layer of the model except the 15-channel RPN head): straight-line,
    // buffer stores with out-of-range offsets for the masked chunks.  The relu flag is
    // unswitched so that the pass bodies have no control flow (the waitcnt placement stays
    // exact: one vmcnt wait for the residual tile, none between the stores).
    auto run = [&](auto relu_c) {
      constexpr bool RELU = decltype(relu_c)::value;
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        stage_pass(pass);
        if (pass == 0) stamp(3);
#pragma unroll
        for (int s2 = 0; s2 < NCH; ++s2) {
          const int m = m0 + pass * RP + row0 + s2 * (256 / C4);
          const bool ok = col_ok && m < M;
          unsigned opix, rpix;
          out_pix(m, ok, opix, rpix);
          const unsigned ooff = (ok && !(dbg & 64)) ? (opix * p.out_ldc + col) * 4u : kOOB;
          f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[(row0 + s2 * (256 / C4)) * CS + c4 * 4]);
          v += bias4;
          v += rv[pass][s2];
          if (RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)ooff, 0, 0);
        }
        if (pass == 0) stamp(4);
      }
    };
    if (p.relu) run(std::true_type{}); else run(std::false_type{});
  } else {
    // ---- generic path (Cout % 4 != 0 or unaligned pixel strides): scalar tails
    for (int pass = 0; pass < PASSES; ++pass) {
      stage_pass(pass);
      for (int s2 = 0; s2 < NCH; ++s2) {
        const int rl = row0 + s2 * (256 / C4);
        const int m = m0 + pass * RP + rl;
        int nv = p.Cout - col;
        nv = nv > 4 ? 4 : nv;
        if (m >= M || nv <= 0) continue;
        size_t opix, rpix = 0;
        if (dense_io) {
          opix = (size_t)m;
          rpix = (size_t)m;
        } else {
          const int n = fast_div(m, p.div_howo_mul, p.div_howo_sh), rr = m - n * HoWo;
          const int ho = fast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
          opix = ((size_t)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox;
          if (p.res_mode != 0)
            rpix = p.res_mode == 1 ? ((size_t)n * p.res_H + ho) * p.res_W + wo
                                   : ((size_t)n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1);
        }
        f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[rl * CS + c4 * 4]);
        v += bias4;
        if (p.res_mode != 0) {
          const float* rp = p.res + rpix * p.res_ldc + col;
          for (int e = 0; e < nv; ++e) v[e] += rp[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        float* op = p.out + opix * p.out_ldc + col;
        for (int e = 0; e < nv; ++e) op[e] = v[e];
      }
    }
  }
  stamp(5);
}

template <int WM, int WN, int TM, int TN, int ST>
void launch_variant(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int M = p.B * p.Ho * p.Wo;
  const unsigned grid = (unsigned)(((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN));
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, ST>), dim3(grid), dim3(256), 0, stream, dev);
}

}  // namespace

static void make_div(unsigned d, unsigned* mul, unsigned* sh) {
  if (d <= 1) { *mul = 0; *sh = 0; return; }
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                 // l = ceil(log2 d) >= 1
  const unsigned pbits = 31 + l;               // floor(n * mul / 2^p) == n / d for all n < 2^31
  *mul = (unsigned)(((1ull << pbits) + d - 1) / d);
  *sh = pbits - 32;
}

void conv_prepare(ConvParams& p) {
  make_div((unsigned)(p.Ho * p.Wo), &p.div_howo_mul, &p.div_howo_sh);
  make_div((unsigned)p.Wo, &p.div_wo_mul, &p.div_wo_sh);
}

double conv_flops(const ConvParams& p) {
  return 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * (double)(p.kh * p.kw * p.Cin);
}

int launch_conv(const ConvParams& p, hipStream_t stream, const ConvParams* dev_params) {
  ODT_CHECK(p.Cin % 32 == 0, "conv: Cin must be a multiple of 32");
  ODT_CHECK(p.in_ldc % 4 == 0, "conv: input pixel stride must be a multiple of 4 floats");
  ODT_CHECK(p.B > 0 && p.Ho > 0 && p.Wo > 0 && p.Cout > 0, "conv: empty problem");
  ODT_CHECK((double)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4.0 < 2147483648.0 &&
            (double)p.Cout * p.kh * p.kw * p.Cin * 4.0 < 2147483648.0,
            "conv: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  ODT_CHECK((double)p.B * p.out_H * p.out_W * p.out_ldc * 4.0 < 2147483648.0 &&
            (p.res_mode == 0 || (double)p.B * p.res_H * p.res_W * p.res_ldc * 4.0 < 2147483648.0),
            "conv: output / residual tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  const long M = (long)p.B * p.Ho * p.Wo;
  const long tiles128 = ((M + 127) / 128) * ((p.Cout + 127) / 128);
  int tile = 0;   // 0 auto | 1: 128x64 | 2: 64x64 | 3: 128x128  (ODT_CONV_TILE: tuning / test knob)
  if (const char* e = getenv("ODT_CONV_TILE")) tile = atoi(e);
  ConvParams q = p;
  conv_prepare(q);
  bool modified = false;
  if (const char* e = getenv("ODT_CONV_DEBUG")) {
    if (atoi(e) != 0) { q.debug = atoi(e); modified = true; }
  }
  // stand-alone calls (tests, tuning) and debug overrides: stage the record in a temporary
  ConvParams* tmp = nullptr;
  if (dev_params == nullptr || modified) {
    ODT_HIP(hipMalloc((void**)&tmp, sizeof(ConvParams)));
    ODT_HIP(hipMemcpy(tmp, &q, sizeof(ConvParams), hipMemcpyHostToDevice));
    dev_params = tmp;
  }
  if (tile == 0) tile = p.Cout <= 64 ? 1 : (tiles128 < 384 ? 2 : 3);
  // LDS stages: the single-stage / 3-workgroups-per-CU variant wins everywhere (measured per
  // layer, profiles/) except the long 1x1 reductions on the 128x128 tile (res4 conv1, K = 1024:
  // every slice is fresh HBM data, the two-slice register+LDS prefetch of ST = 2 hides it better).
  int stages = (tile == 3 && p.kh * p.kw == 1 && p.Cin >= 1024) ? 2 : 1;
  if (const char* e = getenv("ODT_CONV_STAGES")) {     // tuning knob: force 1 or 2
    if (atoi(e) == 1 || atoi(e) == 2) stages = atoi(e);
  }
  if (stages == 1) {
    if (tile == 1) launch_variant<4, 1, 1, 2, 1>(q, dev_params, stream);
    else if (tile == 2) launch_variant<2, 2, 1, 1, 1>(q, dev_params, stream);
    else launch_variant<2, 2, 2, 2, 1>(q, dev_params, stream);
  } else if (tile == 1) {
    launch_variant<4, 1, 1, 2, 2>(q, dev_params, stream);       // 128 x 64
  } else if (tile == 2) {
    launch_variant<2, 2, 1, 1, 2>(q, dev_params, stream);       // 64 x 64: fill the 256 CUs on small M
  } else {
    launch_variant<2, 2, 2, 2, 2>(q, dev_params, stream);       // 128 x 128
  }
  ODT_HIP(hipGetLastError());
  if (tmp != nullptr) {
    ODT_HIP(hipStreamSynchronize(stream));
    ODT_HIP(hipFree(tmp));
  }
  return 0;
}

}  // namespace odt
