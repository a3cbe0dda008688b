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
template <int WM, int WN, int TM, int TN, int ST, bool FINE>
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
  const int cpt = p.Cin >> 5;               // 32-channel slices per tap
  const int cpt2 = p.in2 != nullptr ? p.Cin2 >> 5 : 0;     // slices of the second A source (1x1 only)
  const int nslices = p.kh * p.kw * cpt + cpt2;
  const int K = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);

  // Branch-free operand fetch: raw buffer loads (SRSRC descriptors built from kernel arguments,
  // i.e. provably wave-uniform); rows that fall into the zero padding / past M / past Cout get an
  // out-of-range offset and the hardware returns zeros.
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in2 != nullptr ? p.in2 : p.in), 0,
      (int)(p.in2 != nullptr ? (unsigned)p.B * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt, 0, (int)((unsigned)p.Cout * K * 4u), 0x00020000);

  // ---- loader role: thread -> (row lr + 32*j, 16-byte column lc)
  const int lc = tid & 7, lr = tid >> 3;
  int a_hw0[RA];               // (hi0 << 16) | (wi0 & 0xffff): input row / column of tap (0,0), 16-bit signed each
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
      a_hw0[j] = 0;
      a_img[j] = ok ? (unsigned)m * p.in_ldc * 4u + lc * 16u : kOOB;
    } else {
      const int mm = ok ? m : 0;
      const int n = fast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = fast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      a_hw0[j] = (int)(((unsigned)(ho * p.stride - p.pad_t) << 16) | ((unsigned)(wo * p.stride - p.pad_l) & 0xffffu));
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
  // tuning ablations (ODT_CONV_DEBUG bits: 1 no global loads, 2 no LDS writes, 4 no barriers, 8 no
  // fragment reads, 16 no MFMAs) exist only in -DODT_CONV_ABLATE builds (tools/ab_build.sh)
#ifdef ODT_CONV_ABLATE
#define ABL(bit) (!(dbg & (bit)))
#else
#define ABL(bit) true
#endif
  // load-stream state (runs up to two slices ahead of the MFMA stream)
  int l_cc = 0, l_kh = 0, l_kw = 0;
  unsigned l_k = 0;             // byte offset of the slice inside a weight row
  unsigned a_row[RA];           // byte offset of this tap's pixel for each row (kOOB if padded)
  auto set_tap = [&](int khh, int kww) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int hi = (a_hw0[j] >> 16) + khh * p.dil, wi = (int)(short)(a_hw0[j] & 0xffff) + kww * p.dil;
      const bool v = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && a_img[j] != kOOB;
      a_row[j] = v ? a_img[j] + (unsigned)(hi * p.in_Wa + wi) * pix_bytes : kOOB;
    }
  };
  set_tap(0, 0);

  f32x4 ra[RA], rb[RB];
  bool l_src2 = false;          // the load stream has moved on to the second A source
  int l_cpt = cpt;
  // second source: output row m reads pixel (n, ho * in2_stride, wo * in2_stride) of in2
  auto set_src2 = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = m0 + lr + 32 * j;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n = fast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = fast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      const unsigned pix = ((unsigned)n * p.in2_Ha + (unsigned)(ho * p.in2_stride)) * p.in2_Wa + (unsigned)(wo * p.in2_stride);
      a_row[j] = ok ? pix * (unsigned)p.in2_ldc * 4u + lc * 16u : kOOB;
    }
  };
  auto advance_stream = [&]() {  // next 32-channel slice of the K stream
    l_k += 128u;
    if (++l_cc == l_cpt) {
      l_cc = 0;
      if (!l_src2) {
        if (++l_kw == p.kw) { l_kw = 0; ++l_kh; }
        if (l_kh == p.kh && cpt2 > 0) {
          l_src2 = true; l_cpt = cpt2;
          set_src2();
        } else {
          set_tap(l_kh, l_kw);  // harmless past the last tap (never loaded)
        }
      }
    }
  };
  auto load_a = [&](int j) {
    return (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)a_row[j], l_cc * 128, 0);
  };
  auto load_slice = [&]() {     // fetch the next slice of the stream into ra / rb
#pragma unroll
    for (int j = 0; j < RA; ++j) ra[j] = load_a(j);
#pragma unroll
    for (int j = 0; j < RB; ++j)
      rb[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_wt, (int)b_off[j], (int)l_k, 0);
    advance_stream();
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
  if constexpr (FINE) {
  // ---- fine-grained interleave.  A wave issues in order, so whatever follows a run of MFMAs
  // (global-load issue + address math, LDS writes, fragment reads) is exposed unless ANOTHER wave
  // on the SIMD has MFMAs ready.  Here every non-MFMA instruction of a slice is placed right
  // after one MFMA (64 pipe cycles each), one or two per MFMA, pinned with sched_barrier fences:
  // a lone workgroup on a CU keeps the matrix pipe busy through the whole slice.
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  constexpr int GM = TM * TN;          // MFMAs per k-step
  auto mfma_at = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], int idx) {
    const int t = idx / GM, ij = idx % GM, i = ij / TN, j = ij % TN;
    if (ABL(16)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[j][t], acc[i][j], 0, 0, 0);
  };
  auto read_one = [&](int buf, int g, int s2, f32x4 (&fa)[TM], f32x4 (&fb)[TN]) {
    if (!ABL(8)) return;
    if (s2 < TM)
      fa[s2] = *reinterpret_cast<const f32x4*>(lds[buf] + (wm * TM * 32 + s2 * 32) * LS + frag_off + g * 8);
    else
      fb[s2 - TM] = *reinterpret_cast<const f32x4*>(lds[buf] + BM * LS + (wn * TN * 32 + (s2 - TM) * 32) * LS +
                                                    frag_off + g * 8);
  };
  auto write_one = [&](int buf, int j) {
    if (!ABL(2)) return;
    if (j < RA) *reinterpret_cast<f32x4*>(&lds[buf][(lr + 32 * j) * LS + lc * 4]) = ra[j];
    else *reinterpret_cast<f32x4*>(&lds[buf][BM * LS + (lr + 32 * (j - RA)) * LS + lc * 4]) = rb[j - RA];
  };
  auto gload_one = [&](int j) {        // j == RA + RB: advance the load stream to the next slice
    if (!ABL(1)) return;
    if (j < RA) {
      ra[j] = load_a(j);
    } else if (j < RA + RB) {
      rb[j - RA] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_wt, (int)b_off[j - RA], (int)l_k, 0);
    } else {
      advance_stream();
    }
  };
  // MFMAs [i0, i1) of a group, with side(0..nside-1) spread evenly behind them
  auto run = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN], auto I0, auto I1, auto NS, auto&& side) {
    constexpr int i0 = decltype(I0)::value, n = decltype(I1)::value - i0, nside = decltype(NS)::value;
    constexpr int per = (nside + n - 1) / n;
#pragma unroll
    for (int m = 0; m < n; ++m) {
      mfma_at(fa, fb, i0 + m);
      ODT_FENCE();
#pragma unroll
      for (int q = 0; q < per; ++q)
        if (m * per + q < nside) { side(m * per + q); ODT_FENCE(); }
    }
  };
  using Z = std::integral_constant<int, 0>;
  using NF_ = std::integral_constant<int, TM + TN>;
  using NL_ = std::integral_constant<int, RA + RB>;
  using NA_ = std::integral_constant<int, TM + TN + RA + RB + 1>;
  using G4_ = std::integral_constant<int, 4 * GM>;
  using G2_ = std::integral_constant<int, 2 * GM>;
  constexpr int NF = TM + TN;
  using T_ = std::true_type;
  using F_ = std::false_type;
  // One K slice.  NEXT: slice c+1 exists (write it to LDS, read its first fragments); PRE: slice
  // c+2 exists (fetch it).  The K loop is peeled into (all but two) x full slice, one without
  // the fetch, one without anything behind it, so MFMAs never sit inside a conditional arm (that costs a second accumulator copy).
  auto slice = [&](int c, auto NEXT, auto NW, auto NG) {
    constexpr bool next = decltype(NEXT)::value;
    using NW_ = decltype(NW);           // LDS writes of slice c+1 (0 on the last slice)
    using NG_ = decltype(NG);           // fragment reads of slice c+1 + global fetch of slice c+2
    if constexpr (ST == 2) {
      // stage c&1 holds slice c; registers -> other stage behind the second half of g2; barrier;
      // first fragments of slice c+1 and the global prefetch of slice c+2 behind g3.
      const int cur = c & 1;
      run(fa0, fb0, Z{}, G4_{}, NF_{}, [&](int q) { read_one(cur, 1, q, fa1, fb1); });
      run(fa1, fb1, Z{}, G4_{}, NF_{}, [&](int q) { read_one(cur, 2, q, fa0, fb0); });
      run(fa0, fb0, Z{}, G2_{}, NF_{}, [&](int q) { read_one(cur, 3, q, fa1, fb1); });
      run(fa0, fb0, G2_{}, G4_{}, NW_{}, [&](int q) { write_one(cur ^ 1, q); });
      if (ABL(4)) __syncthreads();
      ODT_FENCE();
      run(fa1, fb1, Z{}, G4_{}, NG_{}, [&](int q) {
        if (next && q < NF) read_one(cur ^ 1, 0, q, fa0, fb0); else gload_one(q - (next ? NF : 0));
      });
    } else {
      // single LDS stage: barrier after the last fragment read of slice c, registers -> LDS behind
      // the first half of g3, barrier, first fragments of slice c+1 + global prefetch of slice
      // c+2 behind the second half.
      run(fa0, fb0, Z{}, G4_{}, NF_{}, [&](int q) { read_one(0, 1, q, fa1, fb1); });
      run(fa1, fb1, Z{}, G4_{}, NF_{}, [&](int q) { read_one(0, 2, q, fa0, fb0); });
      run(fa0, fb0, Z{}, G4_{}, NF_{}, [&](int q) { read_one(0, 3, q, fa1, fb1); });
      if (ABL(4)) __syncthreads();
      ODT_FENCE();
      run(fa1, fb1, Z{}, G2_{}, NW_{}, [&](int q) { write_one(0, q); });
      if (ABL(4)) __syncthreads();
      ODT_FENCE();
      run(fa1, fb1, G2_{}, G4_{}, NG_{}, [&](int q) {
        if (next && q < NF) read_one(0, 0, q, fa0, fb0); else gload_one(q - (next ? NF : 0));
      });
    }
  };
  {
    int c = 0;
    for (; c + 2 < nslices; ++c) slice(c, T_{}, NL_{}, NA_{});
    if (c + 1 < nslices) { slice(c, T_{}, NL_{}, NF_{}); ++c; }
    slice(c, F_{}, Z{}, Z{});
  }
  } else {
  if constexpr (ST == 2) {
    for (int c = 0; c < nslices; ++c) {
      const int cur = c & 1;
      const bool more = c + 1 < nslices;
      if (ABL(8)) read_frags(cur, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(8)) read_frags(cur, 2, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(8)) read_frags(cur, 3, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa0, fb0, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (more && ABL(2)) store_slice(cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa0, fb0, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && ABL(8)) read_frags(cur ^ 1, 0, fa0, fb0);
      if (c + 2 < nslices && ABL(1)) load_slice();
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    // single LDS stage: g0..g2 as above; then barrier (all fragment reads of this slice done),
    // registers -> LDS (slice c+1) with the g3 MFMAs covering the write, fetch slice c+2,
    // barrier, first fragment read of the next slice.
    for (int c = 0; c < nslices; ++c) {
      const bool more = c + 1 < nslices;
      if (ABL(8)) read_frags(0, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(8)) read_frags(0, 2, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa1, fb1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(8)) read_frags(0, 3, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa0, fb0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && ABL(2)) store_slice(0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa1, fb1, 0, 2);
      if (c + 2 < nslices && ABL(1)) load_slice();
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(4)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (more && ABL(8)) read_frags(0, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      if (ABL(16)) mfma_group(fa1, fb1, 2, 4);    // covers the first fragment read of slice c+1
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
#undef ODT_FENCE
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
    auto run = [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
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
          if (ACT == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (ACT == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
          } else if (ACT == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
          }
          __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)ooff, 0, 0);
        }
        if (pass == 0) stamp(4);
      }
    };
    if (p.relu == 1) run(std::integral_constant<int, 1>{});
    else if (p.relu == 2) run(std::integral_constant<int, 2>{});
    else if (p.relu == 3) run(std::integral_constant<int, 3>{});
    else run(std::integral_constant<int, 0>{});
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p.relu == 1) v[e] = fmaxf(v[e], 0.f);
          else if (p.relu == 2) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
          else if (p.relu == 3) v[e] = 1.0f / (1.0f + expf(-v[e]));
        }
        float* op = p.out + opix * p.out_ldc + col;
        for (int e = 0; e < nv; ++e) op[e] = v[e];
      }
    }
  }
  stamp(5);
}

template <int WM, int WN, int TM, int TN, int ST, bool FINE>
void launch_variant(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int M = p.B * p.Ho * p.Wo;
  const unsigned grid = (unsigned)(((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN));
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN, ST, FINE>), dim3(grid), dim3(256), 0, stream, dev);
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
  return 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout *
             (double)(p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0)) +
         (p.head_wt != nullptr ? 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * 15.0 : 0.0) +   // fused 1x1 head (15 real columns)
         (p.f_wt != nullptr ? 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * (double)p.f_cout : 0.0);   // fused 1x1 conv
}

int launch_conv(const ConvParams& p, hipStream_t stream, const ConvParams* dev_params) {
  ODT_CHECK(p.Cin % 32 == 0, "conv: Cin must be a multiple of 32");
  ODT_CHECK(p.in_ldc % 4 == 0, "conv: input pixel stride must be a multiple of 4 floats");
  ODT_CHECK(p.B > 0 && p.Ho > 0 && p.Wo > 0 && p.Cout > 0, "conv: empty problem");
  ODT_CHECK(p.in2 == nullptr || (p.kh == 1 && p.kw == 1 && p.Cin2 % 32 == 0 && p.in2_ldc % 4 == 0 &&
                                 (double)p.B * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4.0 < 2147483648.0),
            "conv: a second A source needs a 1x1 conv, Cin2 % 32 == 0 and a tensor below 2 GiB");
  ODT_CHECK((long)p.Ho * p.stride + (long)p.kh * p.dil < 32000 && (long)p.Wo * p.stride + (long)p.kw * p.dil < 32000 &&
            p.pad_t < 32000 && p.pad_l < 32000, "conv: spatial extent must fit 16-bit tap coordinates");
  ODT_CHECK((double)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4.0 < 2147483648.0 &&
            (double)p.Cout * (p.kh * p.kw * p.Cin + p.Cin2) * 4.0 < 2147483648.0,
            "conv: operand tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  ODT_CHECK((double)p.B * p.out_H * p.out_W * p.out_ldc * 4.0 < 2147483648.0 &&
            (p.res_mode == 0 || (double)p.B * p.res_H * p.res_W * p.res_ldc * 4.0 < 2147483648.0),
            "conv: output / residual tensors must be smaller than 2 GiB (32-bit buffer offsets)");
  ODT_CHECK(p.nlvl <= 1 || (p.wt_split != nullptr && p.wt_split_kind == 3 && p.splitk <= 1 && p.nlvl <= 5 && p.head_wt == nullptr),
            "conv: per-row-range epilogue constants need a conv_split3 kernel without split-K");
  const long M = (long)p.B * p.Ho * p.Wo;
  const long tiles128 = ((M + 127) / 128) * ((p.Cout + 127) / 128);
  int tile = 0;   // 0 auto | 1: 128x64 | 2: 64x64 | 3: 128x128  (ODT_CONV_TILE: tuning / test knob)
  tile = (int)env_knob_long(K_CONV_TILE, 0);
  ConvParams q = p;
  conv_prepare(q);
  bool modified = false;
  if (env_knob(K_CONV_DEBUG).i != 0) { q.debug = (int)env_knob(K_CONV_DEBUG).i; modified = true; }
  // bf16x3 split path: plan convs carry their weight image; stand-alone calls (tests, tuning)
  // build a temporary one
  // (a plan conv without an image stays on the exact-f32 kernel: no allocation on the hot path)
  // (plan convs: the handle's policy decided at plan build -- an attached image means "split"; stand-alone calls:
  // library defaults + ODT_CONV_* overrides, resolved per call)
  struct Temps {                      // temporaries of a stand-alone call: released on every path out of this function
    void* img = nullptr; float* partial = nullptr; unsigned* amax = nullptr; ConvParams* rec = nullptr;
    hipStream_t stream = nullptr; bool used = false;
    ~Temps() {
      if (!used) return;
      (void)hipStreamSynchronize(stream);
      if (rec) (void)hipFree(rec);
      if (img) (void)hipFree(img);
      if (partial) (void)hipFree(partial);
      if (amax) (void)hipFree(amax);
    }
  } tmp;
  tmp.stream = stream;
  ConvPolicy pol{};
  if (q.wt_split == nullptr && dev_params == nullptr) pol = conv_policy_from_env(conv_policy_default());
  const bool split = q.wt_split != nullptr ? true : (dev_params == nullptr && conv_split_wanted(q, pol));
  if (split && q.wt_split == nullptr) {
    const int Ksp = q.kh * q.kw * q.Cin + (q.in2 != nullptr ? q.Cin2 : 0);
    tmp.used = true;
    ODT_HIP(hipMalloc(&tmp.img, conv_split_weight_bytes(q.Cout, Ksp)));
    if (pol.family == 2 && q.in_amax == nullptr) {      // fp16x2 pieces need the sources' |max|: nobody recorded it for a stand-alone call
      // (the scan covers the whole allocation B x in_Ha x in_Wa x in_ldc of a source: a stand-alone caller hands over
      // dense tensors -- odt_op_conv2d* -- so this is the logical view; a sliced view would have to bring its own range)
      ODT_HIP(hipMalloc((void**)&tmp.amax, 2 * kAmaxWays * sizeof(unsigned)));
      ODT_HIP(hipMemsetAsync(tmp.amax, 0, 2 * kAmaxWays * sizeof(unsigned), stream));
      if (launch_tensor_amax(q.in, (size_t)q.B * q.in_Ha * q.in_Wa * q.in_ldc, tmp.amax, stream)) return 1;
      q.in_amax = tmp.amax;
      if (q.in2 != nullptr) {
        if (launch_tensor_amax(q.in2, (size_t)q.B * q.in2_Ha * q.in2_Wa * q.in2_ldc, tmp.amax + kAmaxWays, stream)) return 1;
        q.in2_amax = tmp.amax + kAmaxWays;
      }
    }
    conv_split_choose(q, pol);
    if (conv_make_split_weights(q, tmp.img, stream)) return 1;
    if (q.wt_split_kind == 2) q.h2_chinv = conv_h2_chinv(tmp.img, q.Cout, Ksp);
    q.wt_split = tmp.img; modified = true;
    if (conv_split_partial_bytes(q) > 0) ODT_HIP(hipMalloc((void**)&tmp.partial, conv_split_partial_bytes(q)));
    q.partial = tmp.partial;
  }
  // stand-alone calls (tests, tuning) and debug overrides: stage the record in a temporary
  if (dev_params == nullptr || modified) {
    tmp.used = true;
    ODT_HIP(hipMalloc((void**)&tmp.rec, sizeof(ConvParams)));
    ODT_HIP(hipMemcpy(tmp.rec, &q, sizeof(ConvParams), hipMemcpyHostToDevice));
    dev_params = tmp.rec;
  }
  if (split) return launch_conv_split(q, dev_params, stream);
  // short reductions (K <= 384: EfficientNet / BiFPN 1x1 convs, the res2 / res3 1x1 layers): the
  // 64x64 tile wins -- more workgroups per CU hide the per-tile prologue / epilogue that a two-to-
  // twelve-slice main loop cannot amortise (measured per layer; ODT_CONV_SMALLK=0 for the A/B)
  const bool smallk = !env_knob_off(K_CONV_SMALLK);
  const int Kfull = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  if (tile == 0) tile = p.Cout <= 64 ? 1 : ((tiles128 < 384 || (smallk && Kfull <= 384)) ? 2 : 3);
  // LDS stages: the single-stage / 3-workgroups-per-CU variant wins everywhere (measured per
  // layer, profiles/) except the long 1x1 reductions on the 128x128 tile (res4 conv1, K = 1024:
  // every slice is fresh HBM data, the two-slice register+LDS prefetch of ST = 2 hides it better).
  int stages = (tile == 3 && p.kh * p.kw == 1 && p.Cin >= 1024) ? 2 : 1;
  if (env_knob(K_CONV_STAGES).i == 1 || env_knob(K_CONV_STAGES).i == 2) stages = (int)env_knob(K_CONV_STAGES).i;     // tuning knob: force 1 or 2
  // Loop style (measured per layer, profiles/r01_conv_fine_vs_coarse*.txt): the fine-grained
  // interleave keeps the matrix pipe of a CU busy when few workgroups share it (single-round
  // launches: everything at b=1) and on long reductions; on short reductions with several rounds
  // the workgroups in prologue / epilogue need the issue slots that a never-stalling main loop
  // takes, and the coarse loop wins.
  const int BMt = tile == 2 ? 64 : 128, BNt = tile == 3 ? 128 : 64;
  const long tiles = ((M + BMt - 1) / BMt) * ((p.Cout + BNt - 1) / BNt);
  const long slots = 256L * (stages == 2 ? 2 : (tile == 3 ? 3 : 4));
  const int Kred = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  bool fine = stages == 2 || (tile != 2 && (tiles <= slots || tile == 1 || (Kred >= 1024 && tiles >= 2 * slots)));
  if (tile == 2) fine = tiles >= 384 && tiles <= slots;
  if (env_knob(K_CONV_FINE).c0 == '0' || env_knob(K_CONV_FINE).c0 == '1') fine = env_knob(K_CONV_FINE).c0 == '1';       // tuning knob: force 0 or 1
  if (stages == 1) {
    if (tile == 1) { if (fine) launch_variant<4, 1, 1, 2, 1, true>(q, dev_params, stream); else launch_variant<4, 1, 1, 2, 1, false>(q, dev_params, stream); }
    else if (tile == 2) { if (fine) launch_variant<2, 2, 1, 1, 1, true>(q, dev_params, stream); else launch_variant<2, 2, 1, 1, 1, false>(q, dev_params, stream); }
    else { if (fine) launch_variant<2, 2, 2, 2, 1, true>(q, dev_params, stream); else launch_variant<2, 2, 2, 2, 1, false>(q, dev_params, stream); }
  } else if (tile == 1) {                                        // 128 x 64
    if (fine) launch_variant<4, 1, 1, 2, 2, true>(q, dev_params, stream); else launch_variant<4, 1, 1, 2, 2, false>(q, dev_params, stream);
  } else if (tile == 2) {                                        // 64 x 64: fill the 256 CUs on small M
    if (fine) launch_variant<2, 2, 1, 1, 2, true>(q, dev_params, stream); else launch_variant<2, 2, 1, 1, 2, false>(q, dev_params, stream);
  } else {                                                       // 128 x 128
    if (fine) launch_variant<2, 2, 2, 2, 2, true>(q, dev_params, stream); else launch_variant<2, 2, 2, 2, 2, false>(q, dev_params, stream);
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
