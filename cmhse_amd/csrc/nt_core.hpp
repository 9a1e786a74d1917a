// nt_core.hpp — exact-fp32 MFMA "NT" tile main loop for gfx950 (CDNA4).
//
// Every contraction on the CMHSE hot path has the same shape: C[m][n] = sum_k A[m][k] * B[n][k]
// with BOTH operands K-contiguous rows (hidden states x weight rows, embeddings x embeddings).
// This header implements that tile loop once; the GRU step, the attention-energy GEMM, the
// similarity/rank kernel and the cosine-sim kernel differ only in how rows are addressed
// (loaders) and in their epilogues.
//
// Machine mapping (MI355X_MICROARCH.md / cdna_hip_programming.md §3):
//   * v_mfma_f32_32x32x2_f32: exact fp32 (bitwise a k-ordered fmaf chain), 64 cycles per issue per
//     SIMD, one A and one B VGPR per lane: lane l supplies A[i = l&31][k = l>>5] and
//     B[k = l>>5][j = l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
//   * the k index inside an MFMA is a free permutation as long as A and B agree, so each lane
//     reads a float4 (ds_read_b128) = 4 consecutive k for its lane-half and feeds 4 MFMAs:
//     within a block of 8 k, half h = l>>5 owns k = 8*kb + 4*h + {0,1,2,3}.
//   * LDS tiles are [row][BK=16] with a row stride of 20 floats (80 B): 16-byte aligned for
//     b128 access and conflict-free for the 16-lane b128 read groups.
//   * 256-thread workgroups = 4 waves arranged 2 (M) x 2 (N); global -> register prefetch of
//     chunk c+1 is issued before the MFMAs of chunk c, LDS is double-buffered, one barrier/chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace cmhse {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 16;       // floats of K per LDS chunk
constexpr int kLdsLd = 20;    // LDS row stride in floats
constexpr int kThreads = 256; // 4 waves

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Branch-free guarded 4-float load of row `p` (never null) at k..k+3 (< klim), split in two:
//   issue_row4    always issues the load (address clamped into the row) — global address space,
//                 so it is a global_load (vmcnt only), never a flat_load;
//   finish_row4   zeroes what lies outside the row / the tile once the data is needed.
// Keeping the two apart (with a scheduling barrier after the issue) holds the prefetch of chunk
// c+1 in flight across all MFMAs of chunk c; hipcc otherwise sinks the loads next to their use
// and waits for them there (cdna_hip_programming.md §5 item 4c, Guideline 15).
typedef const __attribute__((address_space(1))) float* gptr_f32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gptr_f32x4;

// Row bases travel as 64-bit integer addresses: an integer -> address_space(1) pointer cast gives
// global_load, a generic-pointer cast would still be emitted as flat_load (which also counts on
// lgkmcnt and would be waited for together with the LDS fragment reads).
typedef uint64_t rowaddr_t;
__device__ __forceinline__ rowaddr_t row_addr(const float* p) {
  return reinterpret_cast<rowaddr_t>(p);
}

template <bool VEC>
__device__ __forceinline__ float4 issue_row4(rowaddr_t p, int k, int klim) {
  float4 v;
  if (VEC) {
    const int kk = (k < klim) ? k : (klim - 4);
    gptr_f32x4 src = (gptr_f32x4)(p + static_cast<rowaddr_t>(kk) * 4u);
    const f32x4 g = *src;
    v = make_float4(g.x, g.y, g.z, g.w);
  } else {
    const int last = klim - 1;
    gptr_f32 g = (gptr_f32)p;
    v.x = g[(k < last) ? k : last];
    v.y = g[(k + 1 < last) ? k + 1 : last];
    v.z = g[(k + 2 < last) ? k + 2 : last];
    v.w = g[(k + 3 < last) ? k + 3 : last];
  }
  return v;
}

template <bool VEC>
__device__ __forceinline__ float4 finish_row4(float4 v, bool valid, int k, int klim) {
  if (VEC) {
    const bool ok = valid && (k < klim);
    v.x = ok ? v.x : 0.f;
    v.y = ok ? v.y : 0.f;
    v.z = ok ? v.z : 0.f;
    v.w = ok ? v.w : 0.f;
  } else {
    v.x = (valid && k < klim) ? v.x : 0.f;
    v.y = (valid && k + 1 < klim) ? v.y : 0.f;
    v.z = (valid && k + 2 < klim) ? v.z : 0.f;
    v.w = (valid && k + 3 < klim) ? v.w : 0.f;
  }
  return v;
}

template <int BM, int BNR>
struct TileSmem {
  static constexpr int kAFloats = BM * kLdsLd;
  static constexpr int kBFloats = BNR * kLdsLd;
  static constexpr int kFloats = 2 * (kAFloats + kBFloats);
  static constexpr size_t kBytes = sizeof(float) * kFloats;
  __device__ static float* a(float* base, int buf) { return base + buf * kAFloats; }
  __device__ static float* b(float* base, int buf) {
    return base + 2 * kAFloats + buf * kBFloats;
  }
};

// One phase of the K loop.
//   BM, BNR      rows of the A / B tile staged per chunk
//   MSUB, NSUB   32-row sub-tiles per wave in M / N
//   NACC         accumulators per M sub-tile the caller owns; sub-tile ns of this phase adds into
//                accumulator (ns == NSUB-1 ? LAST : ns) — lets the GRU share r/z accumulators
//                between its x phase and its h phase while keeping the two n-gate terms apart.
//   arow[i]/brow[i]  per-thread row base pointers (always dereferenceable; aval/bval = false
//                makes the row read as zeros) for the rows this thread stages:
//                row = (tid >> 2) + 64*i, 16-byte slot = tid & 3.
//   VEC          K % 4 == 0: dwordx4 global loads.
template <int BM, int BNR, int MSUB, int NSUB, int NACC, int LAST, bool VEC>
__device__ __forceinline__ void nt_phase(float* smem, const rowaddr_t (&arow)[BM / 64],
                                         const bool (&aval)[BM / 64],
                                         const rowaddr_t (&brow)[BNR / 64],
                                         const bool (&bval)[BNR / 64], int K, int a_row0,
                                         const int (&b_row0)[NSUB], f32x16 (&acc)[MSUB][NACC]) {
  using SM = TileSmem<BM, BNR>;
  constexpr int AP = BM / 64, BP = BNR / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int srow = tid >> 2;
  const int sk = (tid & 3) * 4;
  const int nchunks = (K + kBK - 1) / kBK;
  if (nchunks == 0) return;

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 4;
  float4 ra[AP], rb[BP];
  float4 f0a[MSUB], f0b[NSUB], f1a[MSUB], f1b[NSUB];

  // Pieces of the pipeline (all force-inlined lambdas; `buf` is wave-uniform).
  // Running row pointers (VEC): pa/pb address k = kp + sk of every staged row, so the caller's row
  // bases are dead after this point and the steady-state loop needs no address arithmetic.
  rowaddr_t pa[AP], pb[BP];
  int kp = 0;
#pragma unroll
  for (int i = 0; i < AP; ++i) pa[i] = arow[i] + static_cast<rowaddr_t>(sk) * 4u;
#pragma unroll
  for (int i = 0; i < BP; ++i) pb[i] = brow[i] + static_cast<rowaddr_t>(sk) * 4u;
  auto issue_global = [&](int kn) {
    if (VEC) {
      // tail clamp of issue_row4, relative to the running pointers
      const int kk = (kn < K) ? kn : (K - 4);
      const long long off = static_cast<long long>(kk - (kp + sk)) * 4;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const f32x4 g = *(gptr_f32x4)(pa[i] + static_cast<rowaddr_t>(off));
        ra[i] = make_float4(g.x, g.y, g.z, g.w);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const f32x4 g = *(gptr_f32x4)(pb[i] + static_cast<rowaddr_t>(off));
        rb[i] = make_float4(g.x, g.y, g.z, g.w);
      }
    } else {
#pragma unroll
      for (int i = 0; i < AP; ++i) ra[i] = issue_row4<VEC>(arow[i], kn, K);
#pragma unroll
      for (int i = 0; i < BP; ++i) rb[i] = issue_row4<VEC>(brow[i], kn, K);
    }
  };
  auto write_lds = [&](int buf, int kn) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      *reinterpret_cast<float4*>(SM::a(smem, buf) + (srow + 64 * i) * kLdsLd + sk) =
          finish_row4<VEC>(ra[i], aval[i], kn, K);
#pragma unroll
    for (int i = 0; i < BP; ++i)
      *reinterpret_cast<float4*>(SM::b(smem, buf) + (srow + 64 * i) * kLdsLd + sk) =
          finish_row4<VEC>(rb[i], bval[i], kn, K);
  };
  auto read_frags = [&](int buf, int kb, float4(&fa)[MSUB], float4(&fb)[NSUB]) {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
      fa[ms] = *reinterpret_cast<const float4*>(SM::a(smem, buf) +
                                                (a_row0 + ms * 32 + frow) * kLdsLd + kb * 8 + fk);
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns)
      fb[ns] = *reinterpret_cast<const float4*>(SM::b(smem, buf) +
                                                (b_row0[ns] + frow) * kLdsLd + kb * 8 + fk);
  };
  // MFMAs of one 8-k block for k sub-steps j in [J0, J1)
  auto mfma_block = [&](const float4(&fa)[MSUB], const float4(&fb)[NSUB], auto j0c, auto j1c) {
    constexpr int J0 = decltype(j0c)::value, J1 = decltype(j1c)::value;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
#pragma unroll
      for (int ms = 0; ms < MSUB; ++ms) {
        const float av = (j == 0) ? fa[ms].x : (j == 1) ? fa[ms].y : (j == 2) ? fa[ms].z
                                                                              : fa[ms].w;
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
          const float bv = (j == 0) ? fb[ns].x : (j == 1) ? fb[ns].y : (j == 2) ? fb[ns].z
                                                                                : fb[ns].w;
          constexpr int kLast = LAST;
          const int ai = (ns == NSUB - 1) ? kLast : ns;
          acc[ms][ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[ms][ai], 0, 0, 0);
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I3 = std::integral_constant<int, 3>;
  using I4 = std::integral_constant<int, 4>;

  // Rotated software pipeline.  An MFMA runs for 64 cycles after issue, so a wave's own memory
  // instructions hide under its MFMAs when they are issued just ahead of a block of them:
  //   barrier                      chunk c complete in LDS, every wave holds its (c-1, kb1) frags
  //   LDS reads  F0 <- (c, kb0);  global prefetch R <- chunk c+1
  //   MFMAs (c-1, kb1) from F1     <- covers the LDS-read and global latency
  //   LDS reads  F1 <- (c, kb1)
  //   MFMAs (c, kb0) from F0, k sub-steps 0..2
  //   tail-mask R, LDS writes chunk c+1 -> other buffer (all waves are past reading it)
  //   MFMAs (c, kb0) sub-step 3    <- covers the LDS-write latency before the next barrier
  // Only the barrier itself is exposed.
  issue_global(sk);
  __syncthreads();  // previous phase / kernel section finished with the LDS buffers
  write_lds(0, sk);
  __syncthreads();
  read_frags(0, 0, f0a, f0b);
  issue_global(kBK + sk);
  read_frags(0, 1, f1a, f1b);
  __builtin_amdgcn_sched_barrier(0);
  mfma_block(f0a, f0b, I0{}, I3{});
  __builtin_amdgcn_sched_barrier(0);
  write_lds(1, kBK + sk);
  __builtin_amdgcn_sched_barrier(0);
  mfma_block(f0a, f0b, I3{}, I4{});
  int c = 1;
  if (VEC) {
    // Lean steady state, two chunks per trip.  Every vector-ALU instruction in this loop costs
    // matrix-pipe time (tools/microbench/mfma_lds_feed_f32.hip: 32 of them per 24 MFMAs take 19 %
    // off the loop), so the chunks that lie wholly inside K are staged with no tail clamp, no
    // masks (rows past the tile edge hold a valid row's data and only feed outputs that are never
    // stored) and no per-chunk address arithmetic: the running row pointers advance once per trip,
    // the second chunk's loads use the +64 B immediate, and both LDS buffers are compile-time offsets.
#pragma unroll
    for (int i = 0; i < AP; ++i) pa[i] += static_cast<rowaddr_t>(2 * kBK) * 4u;
#pragma unroll
    for (int i = 0; i < BP; ++i) pb[i] += static_cast<rowaddr_t>(2 * kBK) * 4u;
    kp = 2 * kBK;
    auto lean_chunk = [&](auto curc, auto offc) {
      constexpr int cur = decltype(curc)::value;
      constexpr unsigned off = decltype(offc)::value;
      __syncthreads();
      read_frags(cur, 0, f0a, f0b);
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const f32x4 g = *(gptr_f32x4)(pa[i] + off);
        ra[i] = make_float4(g.x, g.y, g.z, g.w);
      }
#pragma unroll
      for (int i = 0; i < BP; ++i) {
        const f32x4 g = *(gptr_f32x4)(pb[i] + off);
        rb[i] = make_float4(g.x, g.y, g.z, g.w);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f1a, f1b, I0{}, I4{});
      __builtin_amdgcn_sched_barrier(0);
      read_frags(cur, 1, f1a, f1b);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I0{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < AP; ++i)
        *reinterpret_cast<float4*>(SM::a(smem, cur ^ 1) + (srow + 64 * i) * kLdsLd + sk) = ra[i];
#pragma unroll
      for (int i = 0; i < BP; ++i)
        *reinterpret_cast<float4*>(SM::b(smem, cur ^ 1) + (srow + 64 * i) * kLdsLd + sk) = rb[i];
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(f0a, f0b, I3{}, I4{});
    };
    using I1 = std::integral_constant<int, 1>;
    using U0 = std::integral_constant<unsigned, 0u>;
    using U64 = std::integral_constant<unsigned, 64u>;
    // trip (c, c+1), c odd: stages chunks c+1 and c+2, both must end inside K
    for (; (c + 3) * kBK <= K; c += 2) {
      lean_chunk(I1{}, U0{});
      lean_chunk(I0{}, U64{});
#pragma unroll
      for (int i = 0; i < AP; ++i) pa[i] += 128u;
#pragma unroll
      for (int i = 0; i < BP; ++i) pb[i] += 128u;
      kp += 2 * kBK;
    }
  }
  for (; c < nchunks; ++c) {
    const int cur = c & 1;
    const int kn = (c + 1) * kBK + sk;  // past the end: clamped address, zeroed by finish_row4
    __syncthreads();
    read_frags(cur, 0, f0a, f0b);
    issue_global(kn);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f1a, f1b, I0{}, I4{});
    __builtin_amdgcn_sched_barrier(0);
    read_frags(cur, 1, f1a, f1b);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f0a, f0b, I0{}, I3{});
    __builtin_amdgcn_sched_barrier(0);
    write_lds(cur ^ 1, kn);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(f0a, f0b, I3{}, I4{});
  }
  __builtin_amdgcn_sched_barrier(0);
  mfma_block(f1a, f1b, I0{}, I4{});
  __syncthreads();  // all waves done with LDS before the caller reuses it
}

// ---------------------------------------------------------------------------------------------
// bf16x3 variant of nt_phase ("CMHSE_MATH_BF16X3"): fp32-grade products on the bf16 matrix pipe.
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi,  x_hi = bf16(x), x_lo = bf16(x - x_hi)
// (relative error ~2^-17 per product, fp32 accumulation; measured 1e-6 on the normalised
// embeddings after 80 GRU steps, against the 1e-4 parity bar).  Six v_mfma_f32_32x32x8_bf16_1k
// (32 cycles each, 8 k: mfma_bf16_16k below, and why it is not three v_mfma_f32_32x32x16_bf16) replace
// eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7x less matrix time per chunk; the loop is paced
// by operand staging and the in-register split of A as much as by its MFMAs.
//   * A stays fp32 in HBM and LDS; each lane splits its 8-k fragment in registers (v_cvt_pk).
//   * B (weights) is pre-split once per call by split_bf16x3_kernel into rows of the SAME byte
//     length: per 16-k chunk 32 B of hi (16 bf16) then 32 B of lo, so the global->LDS staging code
//     and the LDS tile shape are those of nt_phase and the fragments are plain ds_read_b128.
//   * one chunk (16 k) = one MFMA k-step; same rotated software pipeline.
// ---------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// One 32 x 32 x 16 bf16 product step: lane l holds A[l & 31][8 (l >> 5) + 0..7] and the matching B.
// gfx950's v_mfma_f32_32x32x16_bf16 does it in one instruction (8 passes) — and is NOT used: while waves
// of a kernel issue the double-rate matrix instructions gfx950 added (32x32x16 bf16/f16, 16x16x32 bf16,
// 32x32x32 i8: 128-bit A/B operands), a v_pk_fma_f32 of ANOTHER wave on the same SIMD now and then loses
// the write of lanes 48-63 of one of its two result registers (tools/microbench/pkfma_lost_update.hip:
// up to 3 % of a bystander's sums wrong; profiles/r05_bf16_mfma_bystander.txt).  The two-instruction
// form on v_mfma_f32_32x32x8_bf16_1k (each lane's first four k, then its last four; the k order inside
// a chunk is free as long as A and B agree) shows no such effect; it costs the bf16x3 pass 17 % (the
// mode is 1.7x the exact path instead of 2.0x).  CMHSE_BF3_MFMA_32X32X16 restores the single
// instruction (experiments only: build.py::audit_isa refuses such a library).
typedef short bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_bf16_16k(bf16x8 a, bf16x8 b, f32x16 c) {
#ifdef CMHSE_BF3_MFMA_32X32X16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  const bf16x4 a0 = {a[0], a[1], a[2], a[3]}, a1 = {a[4], a[5], a[6], a[7]};
  const bf16x4 b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
  c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a0, b0, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a1, b1, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // round-to-nearest-even pair
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void split8(const float4& lo4, const float4& hi4, uint4& h, uint4& l) {
  const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hh[i] = pack_bf16(x[2 * i], x[2 * i + 1]);
    const float fa = __uint_as_float(hh[i] << 16), fb = __uint_as_float(hh[i] & 0xffff0000u);
    ll[i] = pack_bf16(x[2 * i] - fa, x[2 * i + 1] - fb);
  }
  h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// ASPLIT: the A rows are PRE-SPLIT like the weights (rows of split_ld(K) float units, per 16-k chunk
// 16 bf16 hi then 16 bf16 lo; written once by split_rows_kernel for the inputs and by the step
// epilogue for the hidden states): no conversion in the loop, the A fragments are read exactly like
// the B fragments, one chunk ahead of the MFMAs that consume them.
template <int BM, int BNR, int MSUB, int NSUB, int NACC, int LAST, bool ASPLIT = false>
__device__ __forceinline__ void nt_phase_bf3(float* smem, const rowaddr_t (&arow)[BM / 64],
                                             const bool (&aval)[BM / 64],
                                             const rowaddr_t (&brow)[BNR / 64],
                                             const bool (&bval)[BNR / 64], int K, int a_row0,
                                             const int (&b_row0)[NSUB], f32x16 (&acc)[MSUB][NACC]) {
  using SM = TileSmem<BM, BNR>;
  constexpr int AP = BM / 64, BP = BNR / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int srow = tid >> 2;
  const int sk = (tid & 3) * 4;
  const int nchunks = (K + kBK - 1) / kBK;
  if (nchunks == 0) return;
  const int Kp = nchunks * kBK;  // pre-split rows are zero-padded to whole chunks by their producers
  const int Ka = ASPLIT ? Kp : K;
  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;  // this lane-half's 8 k inside the 16-k chunk
  float4 ra[AP], rb[BP];
  float4 xa[MSUB][2];               // raw fp32 A fragments of the chunk just read (!ASPLIT)
  uint4 an[MSUB], aln[MSUB];        // pre-split A fragments of the chunk just read (ASPLIT)
  uint4 ah[MSUB], al[MSUB], bh[NSUB], bl[NSUB];   // operands of the chunk being multiplied

  auto issue_global = [&](int kn) {
#pragma unroll
    for (int i = 0; i < AP; ++i) ra[i] = issue_row4<true>(arow[i], kn, Ka);
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[i] = issue_row4<true>(brow[i], kn, Kp);
  };
  auto write_lds = [&](int buf, int kn) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      *reinterpret_cast<float4*>(SM::a(smem, buf) + (srow + 64 * i) * kLdsLd + sk) =
          finish_row4<true>(ra[i], aval[i], kn, Ka);
#pragma unroll
    for (int i = 0; i < BP; ++i)
      *reinterpret_cast<float4*>(SM::b(smem, buf) + (srow + 64 * i) * kLdsLd + sk) =
          finish_row4<true>(rb[i], bval[i], kn, Kp);
  };
  auto read_a = [&](int buf) {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      if (ASPLIT) {
        const float* p = SM::a(smem, buf) + (a_row0 + ms * 32 + frow) * kLdsLd;
        an[ms] = *reinterpret_cast<const uint4*>(p + (fk >> 1));
        aln[ms] = *reinterpret_cast<const uint4*>(p + 8 + (fk >> 1));
      } else {
        const float* p = SM::a(smem, buf) + (a_row0 + ms * 32 + frow) * kLdsLd + fk;
        xa[ms][0] = *reinterpret_cast<const float4*>(p);
        xa[ms][1] = *reinterpret_cast<const float4*>(p + 4);
      }
    }
  };
  auto read_b = [&](int buf) {
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      // chunk row = [16 bf16 hi | 16 bf16 lo]; this half's 8 k = 16 bytes at fk*2 bytes
      const float* p = SM::b(smem, buf) + (b_row0[ns] + frow) * kLdsLd;
      bh[ns] = *reinterpret_cast<const uint4*>(p + (fk >> 1));
      bl[ns] = *reinterpret_cast<const uint4*>(p + 8 + (fk >> 1));
    }
  };
  auto convert_a = [&]() {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      if (ASPLIT) {
        ah[ms] = an[ms];
        al[ms] = aln[ms];
      } else {
        split8(xa[ms][0], xa[ms][1], ah[ms], al[ms]);
      }
    }
  };
  auto mfma_chunk = [&]() {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah[ms]);
      const bf16x8 a_l = __builtin_bit_cast(bf16x8, al[ms]);
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh[ns]);
        const bf16x8 b_l = __builtin_bit_cast(bf16x8, bl[ns]);
        constexpr int kLast = LAST;
        const int ai = (ns == NSUB - 1) ? kLast : ns;
        acc[ms][ai] = mfma_bf16_16k(a_l, b_h, acc[ms][ai]);
        acc[ms][ai] = mfma_bf16_16k(a_h, b_l, acc[ms][ai]);
        acc[ms][ai] = mfma_bf16_16k(a_h, b_h, acc[ms][ai]);
      }
    }
  };

  // prologue: chunk 0 into buffer 0, its fragments into registers, chunk 1 on its way
  issue_global(sk);
  __syncthreads();
  write_lds(0, sk);
  __syncthreads();
  read_a(0);
  read_b(0);
  issue_global(kBK + sk);
  convert_a();
  __builtin_amdgcn_sched_barrier(0);
  write_lds(1, kBK + sk);
  // (a lean two-chunk steady state like nt_phase's measured 5-8 % SLOWER here: this loop is paced by
  // the operand conversion, and the extra live pointers cost registers)
  for (int c = 1; c < nchunks; ++c) {
    const int cur = c & 1;
    const int kn = (c + 1) * kBK + sk;
    __syncthreads();                 // chunk c complete in LDS; everyone holds chunk c-1 operands
    read_a(cur);                     // A of chunk c (into xa / an; ah/al still hold chunk c-1)
    issue_global(kn);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk();                    // chunk c-1
    __builtin_amdgcn_sched_barrier(0);
    read_b(cur);                     // B operands of chunk c (bh/bl are free once the MFMAs issued)
    convert_a();
    __builtin_amdgcn_sched_barrier(0);
    write_lds(cur ^ 1, kn);
  }
  __builtin_amdgcn_sched_barrier(0);
  mfma_chunk();                      // last chunk
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// bf16x3 on PRE-SPLIT operands through an LDS-DMA ring (round 6; the loop of the bf16x3 step kernel and
// of the attention projection of pre-split rows).
//
// tools/tile_trace.py arms of nt_phase_bf3 (profiles/r06_fast_mode.txt): of the 292 us a pair of
// co-resident level-1 tiles spends in its K loops, 62 go with the global loads and 20 more with the
// ds_write pass; the MFMAs alone need 210.  The register-staged prefetch runs ONE chunk ahead (1.2 us at a
// third of the fp32 loop's matrix time per chunk) and there are no registers for a second (230 VGPRs).
// With the ring the same pair needs 259 us, and its own arms show nothing left to wait for (no vmcnt
// wait, no barrier, L2-resident sources: all the same time): what remains above the matrix floor is the
// operand traffic itself.  Here the operands go global -> LDS directly
// (global_load_lds_dwordx4: no VGPR destination, no ds_write) into a ring of THREE stages, so the
// load of chunk c + 2 is issued before the MFMAs of chunk c - 1 and is waited for two chunks later
// with a counted vmcnt — never 0 inside the loop — and a raw s_barrier (a __syncthreads() would
// drain the DMAs: cdna_hip_programming.md section 5, Pipelining across barriers).
//
// LDS image of a stage: [BM A rows | BNR B rows] x 64 B (one 16-k chunk of a pre-split row: 16 bf16
// hi | 16 bf16 lo), row stride 64 B, NO padding — one wave-instruction writes 1 KiB = 16 whole rows
// (lane i -> row i >> 2, 16-byte slot i & 3).  Bank conflicts of the fragment reads are removed by an
// XOR swizzle instead: slot (q ^ ((row >> 2) & 3)) of a row holds its piece q, applied to the per-lane
// SOURCE address of the DMA and to the ds_read_b128 address (rule 21: both sides).  The 16 lanes of a
// ds_read_b128 group (rows {0-3, 12-15, 20-27} or {4-11, 16-19, 28-31} of a 32-row sub-tile, one
// piece) then cover the 64 banks exactly once.
// Rows past the tile's edge are not masked: their base is clamped to a valid row by the caller and
// they only feed outputs that are never stored; K needs no tail — pre-split rows are zero-padded to
// whole chunks by their producers.
// ---------------------------------------------------------------------------------------------
template <int BM, int BNR>
struct RingSmem {
  static constexpr int kStages = 3;
  static constexpr int kStageBytes = (BM + BNR) * 64;
  static constexpr size_t kBytes = static_cast<size_t>(kStages) * kStageBytes;
};

typedef __attribute__((address_space(3))) char* lds_ptr_t;

// N x (16 bytes per lane, global -> LDS): piece i of the wave goes to LDS byte address dst + 4096 i +
// 16 lane (dst wave-uniform) from this lane's source address src[i] + off.  Inline asm on purpose: hipcc
// puts an s_waitcnt vmcnt(0) in front of every ds_read that follows a __builtin_amdgcn_global_load_lds
// it knows to be in flight, which is exactly the wait this ring exists to avoid; these loads are
// invisible to its bookkeeping (the callers count them: vmcnt(N) before the barrier).  M0 (the LDS
// destination base) is compiler-reserved: saved and restored inside the statement; s_nop 0 = the wait
// state between an SALU write of M0 and the LDS-DMA that reads it.
#define CMHSE_GLDS_FIRST_(D, S) "s_mov_b32 m0, " D "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " S ", off\n\t"
#define CMHSE_GLDS_NEXT_(D, S, OFF) "s_add_i32 m0, " D ", " OFF "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 " S ", off\n\t"
template <int N>
__device__ __forceinline__ void glds16_pieces(const rowaddr_t (&src)[N], rowaddr_t off, unsigned dst) {
  static_assert(N == 5 || N == 6, "pieces per wave and chunk");
  unsigned keep;
  const rowaddr_t s0 = src[0] + off, s1 = src[1] + off, s2 = src[2] + off, s3 = src[3] + off, s4 = src[4] + off;
  if constexpr (N == 5) {
    asm volatile("s_mov_b32 %0, m0\n\t" CMHSE_GLDS_FIRST_("%6", "%1") CMHSE_GLDS_NEXT_("%6", "%2", "0x1000")
                 CMHSE_GLDS_NEXT_("%6", "%3", "0x2000") CMHSE_GLDS_NEXT_("%6", "%4", "0x3000")
                 CMHSE_GLDS_NEXT_("%6", "%5", "0x4000") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "s"(dst)
                 : "memory", "scc");
  } else {
    const rowaddr_t s5 = src[5] + off;
    asm volatile("s_mov_b32 %0, m0\n\t" CMHSE_GLDS_FIRST_("%7", "%1") CMHSE_GLDS_NEXT_("%7", "%2", "0x1000")
                 CMHSE_GLDS_NEXT_("%7", "%3", "0x2000") CMHSE_GLDS_NEXT_("%7", "%4", "0x3000")
                 CMHSE_GLDS_NEXT_("%7", "%5", "0x4000") CMHSE_GLDS_NEXT_("%7", "%6", "0x5000") "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4), "v"(s5), "s"(dst)
                 : "memory", "scc");
  }
}
#undef CMHSE_GLDS_FIRST_
#undef CMHSE_GLDS_NEXT_

template <int BM, int BNR, int MSUB, int NSUB, int NACC, int LAST>
__device__ __forceinline__ void nt_phase_bf3_ring(float* smem, const rowaddr_t (&arow)[BM / 64],
                                                  const rowaddr_t (&brow)[BNR / 64], int K, int a_row0,
                                                  const int (&b_row0)[NSUB], f32x16 (&acc)[MSUB][NACC]) {
  using RS = RingSmem<BM, BNR>;
  constexpr int AP = BM / 64, BP = BNR / 64;
  const int nchunks = (K + kBK - 1) / kBK;
  if (nchunks == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* const base = reinterpret_cast<char*>(smem);
  // this lane's 16-byte piece of the rows it stages (row = (tid >> 2) + 64 i, slot tid & 3)
  const unsigned piece = static_cast<unsigned>(((tid & 3) ^ ((tid >> 4) & 3)) * 16);
  // (a stage is [A rows | B rows], 64 rows = 4096 bytes per piece index: pieces 0 .. AP - 1 are A's)
  rowaddr_t src[AP + BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) src[i] = arow[i] + piece;
#pragma unroll
  for (int i = 0; i < BP; ++i) src[AP + i] = brow[i] + piece;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)base)) + wave * 1024u;
  auto dma = [&](int stage, int c) {
    glds16_pieces<AP + BP>(src, static_cast<rowaddr_t>(c) * 64u, lds0 + static_cast<unsigned>(stage * RS::kStageBytes));
  };
  const int frow = lane & 31, half = lane >> 5, sw = (frow >> 2) & 3;
  const int hi_off = frow * 64 + ((half ^ sw) * 16);
  const int lo_off = frow * 64 + (((2 + half) ^ sw) * 16);
  uint4 an[MSUB], aln[MSUB];
  uint4 ah[MSUB], al[MSUB], bh[NSUB], bl[NSUB];
  auto read_a = [&](int stage) {
    const char* sb = base + stage * RS::kStageBytes + a_row0 * 64;
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      an[ms] = *reinterpret_cast<const uint4*>(sb + ms * 2048 + hi_off);
      aln[ms] = *reinterpret_cast<const uint4*>(sb + ms * 2048 + lo_off);
    }
  };
  auto read_b = [&](int stage) {
    const char* sb = base + stage * RS::kStageBytes + BM * 64;
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
      bh[ns] = *reinterpret_cast<const uint4*>(sb + b_row0[ns] * 64 + hi_off);
      bl[ns] = *reinterpret_cast<const uint4*>(sb + b_row0[ns] * 64 + lo_off);
    }
  };
  auto take_a = [&]() {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      ah[ms] = an[ms];
      al[ms] = aln[ms];
    }
  };
  auto mfma_chunk = [&]() {
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      const bf16x8 a_h = __builtin_bit_cast(bf16x8, ah[ms]);
      const bf16x8 a_l = __builtin_bit_cast(bf16x8, al[ms]);
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const bf16x8 b_h = __builtin_bit_cast(bf16x8, bh[ns]);
        const bf16x8 b_l = __builtin_bit_cast(bf16x8, bl[ns]);
        constexpr int kLast = LAST;
        const int ai = (ns == NSUB - 1) ? kLast : ns;
        acc[ms][ai] = mfma_bf16_16k(a_l, b_h, acc[ms][ai]);
        acc[ms][ai] = mfma_bf16_16k(a_h, b_l, acc[ms][ai]);
        acc[ms][ai] = mfma_bf16_16k(a_h, b_h, acc[ms][ai]);
      }
    }
  };
  // chunk c + 1 has landed (this wave's pieces: the AP + BP DMAs of chunk c + 2 may stay in flight;
  // everybody's: the barrier) and this wave's fragment reads of chunk c have returned, so the stage
  // of chunk c may be overwritten after the barrier
#define CMHSE_RING_STEP_()                                        \
  do {                                                            \
    if (AP + BP == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");      \
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                   \
    __builtin_amdgcn_s_barrier();                                 \
  } while (0)
  static_assert(AP + BP == 5 || AP + BP == 6, "counted vmcnt of the ring");
  const int last = nchunks - 1;
  dma(0, 0);
  dma(1, last < 1 ? last : 1);
  CMHSE_RING_STEP_();                  // chunk 0 landed
  read_a(0);
  read_b(0);
  take_a();
  dma(2, last < 2 ? last : 2);
  CMHSE_RING_STEP_();                  // chunk 1 landed
  int stage = 1;                       // stage of chunk c
  for (int c = 1; c < nchunks; ++c) {
    const int prev = (stage == 0) ? 2 : stage - 1;     // chunk c - 1's stage = chunk c + 2's
    read_a(stage);
    dma(prev, (c + 2 < last) ? c + 2 : last);
    __builtin_amdgcn_sched_barrier(0);
    mfma_chunk();                      // chunk c - 1
    __builtin_amdgcn_sched_barrier(0);
    read_b(stage);
    take_a();
    CMHSE_RING_STEP_();
    stage = (stage == 2) ? 0 : stage + 1;
  }
  __builtin_amdgcn_sched_barrier(0);
  mfma_chunk();                        // last chunk
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last chunk
  __builtin_amdgcn_s_barrier();        // ring free for the next phase
#undef CMHSE_RING_STEP_
}

// Split-K building block of the latency-shaped kernels (gru_step_tiny_kernel, gru_bwd_step_kernel):
// one 32x32 accumulator; NW (4 or 8) waves split K, this wave takes the 8-k blocks wave, wave+NW, ...; the A and B
// fragments (row = lane&31, k = 8*kb + 4*(lane>>5) .. +3) go global -> registers directly in MFMA
// layout through a 4-deep register ring, no LDS and no barrier.
constexpr int kTinyRing = 4;

template <bool VEC, int NW = 4>
__device__ __forceinline__ void tiny_phase(rowaddr_t arow, rowaddr_t brow, bool bvalid, int K,
                                           int wave, int hi, f32x16& acc) {
  // NW waves split K: this wave takes the 8-k blocks wave, wave + NW, ...
  const int nkb = (K + 7) / 8;                          // k-blocks of 8 in this phase
  const int nmine = (nkb - wave + NW - 1) / NW;
  if (nmine <= 0) return;
  float4 ra[kTinyRing], rb[kTinyRing];
#pragma unroll
  for (int d = 0; d < kTinyRing; ++d) {
    const int k = (wave + NW * d) * 8 + 4 * hi;
    ra[d] = issue_row4<VEC>(arow, k, K);
    rb[d] = issue_row4<VEC>(brow, k, K);
  }
  int it = 0;
  if (VEC) {
    // Lean steady state (see nt_phase): while every k this trip consumes or prefetches lies inside
    // K — for all NW waves, so the bound is uniform — no tail masks, no clamps, and the loads go
    // through running pointers with immediate offsets (slot d of the ring is 32 NW bytes further).
    // Columns that are never stored (bvalid == false) need no zeroing either.
    constexpr unsigned kSlot = NW * 8u * 4u;               // bytes between ring slots
    constexpr unsigned kAhead = kSlot * kTinyRing;         // prefetch distance in bytes
    rowaddr_t pa = arow + static_cast<rowaddr_t>(wave * 8 + 4 * hi) * 4u + kAhead;
    rowaddr_t pb = brow + static_cast<rowaddr_t>(wave * 8 + 4 * hi) * 4u + kAhead;
    for (; 8 * NW * it + 64 * NW <= K; it += kTinyRing) {
#pragma unroll
      for (int d = 0; d < kTinyRing; ++d) {
        const float4 a = ra[d], b = rb[d];
        const f32x4 ga = *(gptr_f32x4)(pa + kSlot * d);
        const f32x4 gb = *(gptr_f32x4)(pb + kSlot * d);
        ra[d] = make_float4(ga.x, ga.y, ga.z, ga.w);
        rb[d] = make_float4(gb.x, gb.y, gb.z, gb.w);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
      pa += kSlot * kTinyRing;
      pb += kSlot * kTinyRing;
    }
  }
  for (; it < nmine; it += kTinyRing) {
#pragma unroll
    for (int d = 0; d < kTinyRing; ++d) {
      const int k = (wave + NW * (it + d)) * 8 + 4 * hi;
      const float4 a = finish_row4<VEC>(ra[d], true, k, K);
      const float4 b = finish_row4<VEC>(rb[d], bvalid, k, K);
      const int kn = k + NW * kTinyRing * 8;
      ra[d] = issue_row4<VEC>(arow, kn, K);
      rb[d] = issue_row4<VEC>(brow, kn, K);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// mid_phase: the K loop of the mid-size (few-sequence) steps, forward (gru.hip) and BPTT (bwd.hip):
// 16 x 16 x 4 MFMA blocks, operands global -> registers in MFMA layout, K split over NW waves.
// ---------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

// One wave's share of the K = H contraction for MB x 3 blocks of 16 x 16 outputs.  Blocks of 16 k
// are owned in ADJACENT PAIRS (wave w: blocks 2w, 2w+1, then 2w + 2 NW, ...): a lane quarter loads
// 16 bytes, so one load instruction covers 64 contiguous bytes of each of its 16 rows, and the
// pair, issued back to back, the whole 128-byte line.
template <int MB, int NB, int NW, int D>
__device__ __forceinline__ void mid_phase(const rowaddr_t (&arow)[MB], const rowaddr_t (&brow)[NB],
                                          int K, int wave, int kq, f32x4v (&acc)[MB][NB]) {
  static_assert(D % 2 == 0, "ring holds whole block pairs");
  const int nkb = (K + 15) / 16;
  // ring position i of this wave -> block index
  auto block_of = [&](int i) { return (i >> 1) * 2 * NW + 2 * wave + (i & 1); };
  // positions this wave owns: all i with block_of(i) < nkb (monotone in i)
  int nmine = 0;
  while (block_of(nmine) < nkb) ++nmine;
  if (nmine <= 0) return;
  float4 ra[D][MB], rb[D][NB];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d >= nmine) continue;   // (wave-uniform) nothing to fetch: the slot is never consumed
    const int k = block_of(d) * 16 + 4 * kq;
#pragma unroll
    for (int i = 0; i < MB; ++i) ra[d][i] = issue_row4<true>(arow[i], k, K);
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[d][i] = issue_row4<true>(brow[i], k, K);
  }
  auto mfmas = [&](const float4 (&a)[MB], const float4 (&b)[NB]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float av = (j == 0) ? a[mb].x : (j == 1) ? a[mb].y : (j == 2) ? a[mb].z : a[mb].w;
#pragma unroll
        for (int g = 0; g < NB; ++g) {
          const float bv = (j == 0) ? b[g].x : (j == 1) ? b[g].y : (j == 2) ? b[g].z : b[g].w;
          acc[mb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mb][g], 0, 0, 0);
        }
      }
    }
  };
  int it = 0;
  {
    // lean steady state: every block this trip consumes or prefetches lies wholly inside K for all
    // waves (uniform bound): no masks, no clamps, running pointers with immediate offsets.
    // Ring slot d holds block_of(it + d); slots d, d+1 of a pair are 64 bytes apart, pairs 2 NW blocks.
    constexpr unsigned kPair = 2u * NW * 16u * 4u;        // bytes between consecutive pairs
    constexpr unsigned kAhead = kPair * (D / 2);
    rowaddr_t pa[MB], pb[NB];
    const rowaddr_t lane_off = static_cast<rowaddr_t>(2 * wave * 16 + 4 * kq) * 4u + kAhead;
#pragma unroll
    for (int i = 0; i < MB; ++i) pa[i] = arow[i] + lane_off;
#pragma unroll
    for (int i = 0; i < NB; ++i) pb[i] = brow[i] + lane_off;
    for (; 16 * 2 * NW * ((it + 2 * D) / 2) <= K; it += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const unsigned off = kPair * (d >> 1) + 64u * (d & 1);
        float4 a[MB], b[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          a[i] = ra[d][i];
          const f32x4 g = *(gptr_f32x4)(pa[i] + off);
          ra[d][i] = make_float4(g.x, g.y, g.z, g.w);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          b[i] = rb[d][i];
          const f32x4 g = *(gptr_f32x4)(pb[i] + off);
          rb[d][i] = make_float4(g.x, g.y, g.z, g.w);
        }
        mfmas(a, b);
      }
#pragma unroll
      for (int i = 0; i < MB; ++i) pa[i] += kPair * (D / 2);
#pragma unroll
      for (int i = 0; i < NB; ++i) pb[i] += kPair * (D / 2);
    }
  }
  for (; it < nmine; it += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (it + d >= nmine) continue;   // wave-uniform
      const int k = block_of(it + d) * 16 + 4 * kq;
      const int kn = block_of(it + d + D) * 16 + 4 * kq;
      const bool more = it + d + D < nmine;
      float4 a[MB], b[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        a[i] = finish_row4<true>(ra[d][i], true, k, K);
        if (more) ra[d][i] = issue_row4<true>(arow[i], kn, K);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        b[i] = finish_row4<true>(rb[d][i], true, k, K);
        if (more) rb[d][i] = issue_row4<true>(brow[i], kn, K);
      }
      mfmas(a, b);
    }
  }
}


// Row / column owned by accumulator register r of lane `lane` inside a 32x32 sub-tile.
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

}  // namespace cmhse
