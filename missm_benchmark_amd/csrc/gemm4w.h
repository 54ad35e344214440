// 256x128 NT GEMM tile for FOUR waves with TWO workgroups resident per CU (bf16 MFMA 16x16x32) - included by gemm.hip.
//
// Why a third tile kernel: the 8-phase 256x256 kernel (gemm8p.h) owns a CU alone (128 KiB of LDS, 2 x 229 VGPRs per SIMD), so a
// tile's prologue, epilogue and the workgroup relaunch (1-2 + 2.4-7.5 + 2.6-4.6 us, in-kernel stamps) are serial with its
// 16.4 us main loop at K = 768: the matrix pipe idles for a third of every tile.  Here a workgroup is HALF a CU's worth of
// resources - 4 waves (one per SIMD), 80 KiB of LDS, <= 256 VGPRs - so two of them are resident and drift out of phase by
// themselves: while one is in its epilogue or being relaunched the other one has the matrix pipe to itself.
//   * a wave owns the same 128 x 64 outputs as in gemm8p.h (rows 64 wr + {0, 128} + .., columns 64 wc + ..; 128 accumulator
//     VGPRs, identical register picture, so the branch-free epilogue gemm8p_store_tile is shared);
//   * operands travel as 16 KiB half tiles of one K tile (64): B (128 columns), A0 (rows 0-127), A1 (rows 128-255), sequence
//     number s = 3 t + h, through a ring of FIVE slots (slot = s mod 5, a run-time index); the slot of a half tile is free as soon
//     as every wave holds its fragments in registers, and is refilled at once with half tile s + 5: a request is issued two to three
//     phases (1 to 1.5 K tiles) before the wait that needs it - less slack than the 8-phase kernel's 1.75 K tiles, covered by the
//     co-resident workgroup;
//   * a K tile = 2 phases of 32 MFMAs; the fragments of a phase are read one phase ahead, under the previous phase's MFMAs:
//       phase (t, 0): request s+5, s+6 | read A1(t) | MFMAs A0(t) x B(t) | lgkmcnt(0), vmcnt(8) | barrier
//       phase (t, 1): request s+7      | read A0(t+1) | MFMAs A1(t) x B(t) | read B(t+1) | lgkmcnt(0), vmcnt(8) | barrier
//     (s = 3 t).  vmcnt is counted in issue order: 4 loads per half tile per wave, vmcnt(8) leaves the two youngest half tiles
//     in flight; requests past the last K tile are issued against a zero-record descriptor (dropped, but counted), so the
//     counts are the same on every K tile.
// Hazards: RAW - a half tile is read after the barrier that follows the counted wait of EVERY wave that moved a piece of it;
// WAR - the request that overwrites a slot is issued after the barrier that follows the last fragment read of that slot
// (lgkmcnt(0) before the barrier).
#pragma once

namespace missm {

struct Gemm4wSrc {
  __amdgpu_buffer_rsrc_t a[2], b, none;      // A rows of half 0 / 1, B rows (C columns) of the tile, zero-record descriptor
  unsigned va[4], vb[4];                     // per-lane byte offsets of this wave's four 1 KiB pieces inside a half tile
};

// one half tile = 16 pieces of 1 KiB; wave w moves pieces 4 w .. 4 w + 3 (LDS rows 8 * piece .. + 7, 128 bytes of k each)
template <int H>   // 0: B, 1: A0, 2: A1
__device__ __forceinline__ void stage4w(char* lds, int slot, const Gemm4wSrc& s, int wave, int kbyte, bool live) {
  using lptr = __attribute__((address_space(3))) void*;
  const __amdgpu_buffer_rsrc_t r = live ? (H == 0 ? s.b : s.a[H - 1]) : s.none;
  char* dst = lds + slot * 16384 + wave * 4096;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)(dst + q * 1024), 16, H == 0 ? s.vb[q] : s.va[q], kbyte, 0, 0);
}

template <int H>
__device__ __forceinline__ void stage4w_piece(char* lds, int slot, const Gemm4wSrc& s, int wave, int kbyte, bool live, int q) {
  using lptr = __attribute__((address_space(3))) void*;
  const __amdgpu_buffer_rsrc_t r = live ? (H == 0 ? s.b : s.a[H - 1]) : s.none;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)(lds + slot * 16384 + wave * 4096 + q * 1024), 16, H == 0 ? s.vb[q] : s.va[q], kbyte, 0, 0);
}

template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// tile `logical` of the launch: its group's operand pointers (grouped launch), its origin and its buffer descriptors
__device__ __forceinline__ void gemm4w_tile(const GemmArgs& gall, int logical, GemmArgs& g, Gemm4wSrc& src, int& m0, int& n0) {
  int tm, tn;
  tile_of(logical, gall.tiles_m, gall.tiles_n, gall.group_m, tm, tn);
  if (gall.ngroups > 1) {
    const int gi = tm / gall.group_tiles_m;
    tm -= gi * gall.group_tiles_m;
    select_group(g, gall, gi);
  }
  m0 = tm * 256; n0 = tn * 128;
  const bf16* A = static_cast<const bf16*>(g.A);
  const bf16* B = static_cast<const bf16*>(g.B);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ra = g.M - (m0 + 128 * h);
    const unsigned na = ra <= 0 ? 0u : (unsigned)min(ra, 128) * (unsigned)g.lda * 2u;
    src.a[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (size_t)(m0 + 128 * h) * g.lda), 0, na, 0x00020000);
  }
  const int rb = g.N - n0;
  const unsigned nb = rb <= 0 ? 0u : (unsigned)min(rb, 128) * (unsigned)g.ldb * 2u;
  src.b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B + (size_t)n0 * g.ldb), 0, nb, 0x00020000);
  src.none = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, 0, 0x00020000);
}

// half tiles s = 0 .. 4 of a tile (B0 A0_0 A1_0 | B1 A0_1) into slots 0 .. 4
__device__ __forceinline__ void gemm4w_prologue(char* lds, const Gemm4wSrc& src, int wave, int nt) {
  stage4w<0>(lds, 0, src, wave, 0, true);
  stage4w<1>(lds, 1, src, wave, 0, true);
  stage4w<2>(lds, 2, src, wave, 0, true);
  stage4w<0>(lds, 3, src, wave, 128, nt > 1);
  stage4w<1>(lds, 4, src, wave, 128, nt > 1);
}

// PERSISTENT form (gall.sched, round 3): 512 workgroups (two per CU) stay resident and draw their tiles from the per-XCD queues of
// gemm8p.h.  In-kernel stamps of the one-tile-per-workgroup form showed why this kernel only tied with the 8-phase one on the
// K = 768 video shapes although its two workgroups per CU hide each other's epilogue: a tile slot stood empty for 3.5 - 5 us between
// two workgroups (a workgroup ends only when its stores have drained - 34 GB/s per CU - and its successor starts with a cold
// prologue): 3546 QKV tiles x 17.5 us / 512 slots = 121 us of work in a 157 us launch.  Here the next tile's first five half tiles are
// requested before the epilogue and the epilogue's stores stay in flight across the next tile's first counted wait.
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(GemmArgs gall) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 5 slots x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int nt = gall.K >> 6;                // K tiles (host guarantees K % 64 == 0, K >= 128)

  // ---- tile of this workgroup (static: one tile; dynamic: queue entry blockIdx.x >> 3 of XCD queue blockIdx.x & 7 first, see gemm8p.h)
  const bool dyn = gall.sched != nullptr;    // scalar
  const int qx = blockIdx.x & 7;
  const int q_cnt = (ntiles >> 3) + (qx < (ntiles & 7) ? 1 : 0);
  const int q_start = xcd_remap(qx, ntiles);
  const int q_wgs = ((int)gridDim.x >> 3) + (qx < ((int)gridDim.x & 7) ? 1 : 0);
  int logical = xcd_remap((int)blockIdx.x, ntiles);
  GemmArgs g = gall;                         // (scalar fields only are ever read through this copy)
  Gemm4wSrc src;
  int m0, n0;
  gemm4w_tile(gall, logical, g, src, m0, n0);
  // ---- global -> LDS addressing.  LDS row r of a half tile holds 128 bytes of k, chunk c stored at c ^ (r & 7).
  //  A half ha : LDS row r <-> C row    m0 + 128 ha + r                         (wave wr reads rows 64 wr + 16 i + li)
  //  B         : LDS row r = 64 hb + r' <-> C column n0 + 64 (r' >> 5) + 4 (r' & 15) + 2 hb + ((r' >> 4) & 1)
  //              (wave wc reads rows 64 hb + 32 wc + 16 j + li: with both hb a lane owns the 4 CONSECUTIVE columns 64 wc + 4 li + 0..3)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 32 * wave + 8 * q + (lane >> 3);             // LDS row inside the half tile
    const int c = (lane & 7) ^ (r & 7);                        // the k chunk that lands at stored position lane & 7
    src.va[q] = (unsigned)r * (unsigned)gall.lda * 2u + (unsigned)c * 16u;
    const int hb = r >> 6, rp = r & 63;
    const int col = 64 * (rp >> 5) + 4 * (rp & 15) + 2 * hb + ((rp >> 4) & 1);
    src.vb[q] = (unsigned)col * (unsigned)gall.ldb * 2u + (unsigned)c * 16u;
  }
  gemm4w_prologue(lds, src, wave, nt);

  // ---- fragment read addresses (bytes): row * 128 + ((4 ks + lg) ^ (row & 7)) * 16; + slot base and 16-row-tile immediates
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aoff[2], boff[2];                 // [k step]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned sw = (unsigned)(((ks * 4 + lg) ^ (li & 7)) << 4);
    aoff[ks] = lds0 + (unsigned)(wr * 64 + li) * 128u + sw;
    boff[ks] = lds0 + (unsigned)(wc * 32 + li) * 128u + sw;
  }

  int epi_ops = 0;                           // vector-memory operations of the previous tile's epilogue (0: first tile / guarded path)
  for (;;) {                                 // ---- one output tile per iteration
  unsigned long long t_start = 0, t_loop = 0, t_loop_end = 0;  // diagnostic runs only (tools/gemm_timeline.py)
  if (g.dbg) t_start = __builtin_amdgcn_s_memrealtime();
  unsigned drawn = 0;
  if (dyn && tid == 0) drawn = __hip_atomic_fetch_add(gall.sched + qx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const f32x4 bias4 = prefetch_bias(g, n0 + wc * 64, 0, lane);
  f32x4 acc[2][2][4][2];                     // [A half][B half][16-row tile][16-column tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // (starting the workgroup in the odd hardware wave slot half a phase late - the two workgroups of a CU that start together
  //  might run phase-locked - was measured: no effect at 4096^3, +2 % at K = 2304 / 3072, -5..7 % at K = 768; not kept)
  // B0, A0_0, A1_0 have landed (this wave's pieces).  Issue order: [5 half tiles = 20 requests | previous epilogue (epi_ops) | draw,
  // bias]: they are complete once at most the youngest 8 + epi_ops operations are pending (the counter holds 6 bits).
  if (epi_ops == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if (epi_ops == 32) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
  else if (epi_ops == 48) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
  else if (epi_ops == 64) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();                                // ... everybody's
  if (g.dbg) t_loop = __builtin_amdgcn_s_memrealtime();

  bf16x8 fa[2][4][2], fb[2][2][2];           // A fragments, double-buffered [buffer][i][ks]; B [hb][j][ks]

#define MISSM_4W_MFMA(HA, BUF)                                                                \
  __builtin_amdgcn_s_setprio(1);                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int hb = 0; hb < 2; ++hb)                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
          acc[HA][hb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[BUF][i][ks], fb[hb][j][ks], acc[HA][hb][i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);
#define MISSM_4W_READ_A(SLOT, BUF)                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    const unsigned ad = aoff[ks] + (unsigned)(SLOT) * 16384u;                                 \
    fa[BUF][0][ks] = lds_read128<0 * 2048>(ad);                                               \
    fa[BUF][1][ks] = lds_read128<1 * 2048>(ad);                                               \
    fa[BUF][2][ks] = lds_read128<2 * 2048>(ad);                                               \
    fa[BUF][3][ks] = lds_read128<3 * 2048>(ad);                                               \
  }
#define MISSM_4W_READ_B(SLOT)                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    const unsigned bd = boff[ks] + (unsigned)(SLOT) * 16384u;                                 \
    fb[0][0][ks] = lds_read128<0>(bd);                                                        \
    fb[0][1][ks] = lds_read128<2048>(bd);                                                     \
    fb[1][0][ks] = lds_read128<8192>(bd);                                                     \
    fb[1][1][ks] = lds_read128<8192 + 2048>(bd);                                              \
  }
#define MISSM_4W_FENCE_A(BUF)                                                                 \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
               : "+v"(fa[BUF][0][0]), "+v"(fa[BUF][1][0]), "+v"(fa[BUF][2][0]), "+v"(fa[BUF][3][0]), "+v"(fa[BUF][0][1]),       \
                 "+v"(fa[BUF][1][1]), "+v"(fa[BUF][2][1]), "+v"(fa[BUF][3][1]));              \
  __builtin_amdgcn_sched_barrier(0);
#define MISSM_4W_FENCE_AB(BUF)                                                                \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
               : "+v"(fa[BUF][0][0]), "+v"(fa[BUF][1][0]), "+v"(fa[BUF][2][0]), "+v"(fa[BUF][3][0]), "+v"(fa[BUF][0][1]),       \
                 "+v"(fa[BUF][1][1]), "+v"(fa[BUF][2][1]), "+v"(fa[BUF][3][1]), "+v"(fb[0][0][0]), "+v"(fb[0][1][0]),          \
                 "+v"(fb[1][0][0]), "+v"(fb[1][1][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][1]), "+v"(fb[1][0][1]), "+v"(fb[1][1][1])); \
  __builtin_amdgcn_sched_barrier(0);

  // Fragments are read ONE PHASE AHEAD of their MFMAs, and every request / fragment read is issued BETWEEN two MFMAs of the
  // running phase (one side instruction per two MFMAs): a lone wave per SIMD has no partner whose MFMAs would cover its LDS
  // reads and DMA issue (reads -> barrier -> MFMAs kept the matrix pipe 34 % busy per wave), so the wave overlaps them itself.
  // A fragments are double-buffered (244 VGPRs in all); the B fragments of the next K tile follow the last use of the old ones
  // (k step 0 under the second half of phase 1, k step 1 behind it).  One barrier per phase: it publishes the landed half
  // tiles and releases the slots whose reads it follows.
#define MISSM_4W_MF(HA, BUF, N)                                                               \
  {                                                                                           \
    constexpr int ks_ = (N) >> 4, i_ = ((N) >> 2) & 3, hb_ = ((N) >> 1) & 1, j_ = (N) & 1;       \
    acc[HA][hb_][i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[BUF][i_][ks_], fb[hb_][j_][ks_], acc[HA][hb_][i_][j_], 0, 0, 0); \
  }
#define MISSM_4W_RA(SLOT, BUF, R)                                                             \
  {                                                                                           \
    constexpr int ks_ = (R) >> 2, i_ = (R) & 3;                                               \
    fa[BUF][i_][ks_] = lds_read128<i_ * 2048>(aoff[ks_] + (unsigned)(SLOT) * 16384u);         \
  }
#define MISSM_4W_RB(SLOT, R)                                                                  \
  {                                                                                           \
    constexpr int ks_ = (R) >> 2, hb_ = ((R) >> 1) & 1, j_ = (R) & 1;                          \
    fb[hb_][j_][ks_] = lds_read128<hb_ * 8192 + j_ * 2048>(boff[ks_] + (unsigned)(SLOT) * 16384u); \
  }
  int sB = 0, sA0 = 1, sA1 = 2;              // slots of K tile t's half tiles: (3 t + h) mod 5
  MISSM_4W_READ_B(sB)
  MISSM_4W_READ_A(sA0, 0)
  MISSM_4W_FENCE_AB(0)
  __builtin_amdgcn_s_barrier();              // B(0), A0(0) are in registers everywhere: their slots are free
  for (int t = 0; t < nt; ++t) {
    const int kb1 = (t + 1) * 128, kb2 = (t + 2) * 128;
    const bool live1 = t + 1 < nt, live2 = t + 2 < nt;
    const int nB = sB + 3 >= 5 ? sB - 2 : sB + 3, nA0 = sA0 + 3 >= 5 ? sA0 - 2 : sA0 + 3, nA1 = sA1 + 3 >= 5 ? sA1 - 2 : sA1 + 3;
    // ---- phase 0: A half 0 x B.  Side: requests s = 3 t + 5 (A1(t + 1) -> B(t)'s slot), 3 t + 6 (B(t + 2) -> A0(t)'s slot), reads of A1(t)
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 32>([&](auto n_) {
      constexpr int n = decltype(n_)::value;
      MISSM_4W_MF(0, 0, n)
      if constexpr (n % 2 == 1) {
        constexpr int k = n / 2;               // 0 .. 15: request pieces and fragment reads alternate
        if constexpr (k % 2 == 0) {
          if constexpr (k / 2 < 4) stage4w_piece<2>(lds, sB, src, wave, kb1, live1, k / 2);
          else stage4w_piece<0>(lds, sA0, src, wave, kb2, live2, k / 2 - 4);
        } else {
          MISSM_4W_RA(sA1, 1, k / 2)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);
    MISSM_4W_FENCE_A(1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // B(t + 1), A0(t + 1) have landed (3 t + 5, 3 t + 6 stay in flight)
    __builtin_amdgcn_s_barrier();                              // ... everywhere; A1(t)'s slot is free
    // ---- phase 1: A half 1 x B.  Side: request 3 t + 7 (A0(t + 2) -> A1(t)'s slot), reads of A0(t + 1), then of B(t + 1)
    __builtin_amdgcn_s_setprio(1);
    static_for<0, 32>([&](auto n_) {
      constexpr int n = decltype(n_)::value;
      MISSM_4W_MF(1, 1, n)
      if constexpr (n % 2 == 1) {
        constexpr int k = n / 2;               // 0 .. 15
        if constexpr (k < 8) {                 // first half (k step 0 MFMAs): 4 request pieces, 4 A reads
          if constexpr (k % 2 == 0) stage4w_piece<1>(lds, sA1, src, wave, kb2, live2, k / 2);
          else MISSM_4W_RA(nA0, 0, k / 2)
        } else {                               // second half (k step 1 MFMAs): 4 A reads, the 4 B reads of k step 0 (free now)
          if constexpr (k % 2 == 0) MISSM_4W_RA(nA0, 0, 4 + (k - 8) / 2)
          else MISSM_4W_RB(nB, (k - 8) / 2)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);
    MISSM_4W_RB(nB, 4) MISSM_4W_RB(nB, 5) MISSM_4W_RB(nB, 6) MISSM_4W_RB(nB, 7)
    MISSM_4W_FENCE_AB(0)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // A1(t + 1) has landed (3 t + 6, 3 t + 7 stay in flight)
    __builtin_amdgcn_s_barrier();                              // ... everywhere; B(t + 1)'s and A0(t + 1)'s slots are free
    sB = nB; sA0 = nA0; sA1 = nA1;
  }
#undef MISSM_4W_MF
#undef MISSM_4W_RA
#undef MISSM_4W_RB
#undef MISSM_4W_MFMA
#undef MISSM_4W_READ_A
#undef MISSM_4W_READ_B
#undef MISSM_4W_FENCE_A
#undef MISSM_4W_FENCE_AB
  if (g.dbg) t_loop_end = __builtin_amdgcn_s_memrealtime();
  // every fragment read of this tile is complete and everybody is past the last barrier: all five slots are free.  The drawn queue
  // entry travels through the first word of the (idle) ring - there is no LDS byte to spare at two workgroups per CU -, a second
  // barrier keeps the next tile's requests behind everybody's read of it.
  bool more = false;                         // wave-uniform
  int nlogical = 0;
  if (dyn) {
    // (the word lies in wave 0's piece of slot 0: its last - dropped, zero-filling - requests must have landed before it is written)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) *reinterpret_cast<volatile unsigned*>(lds) = drawn;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned d = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(lds));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int k = q_wgs + (int)d;
    more = k < q_cnt;
    nlogical = q_start + k;
    if ((int)d == q_cnt - 1 && tid == 0) __hip_atomic_store(gall.sched + qx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (more) {
      GemmArgs gn = gall;
      Gemm4wSrc srcn = src;
      int m0n, n0n;
      gemm4w_tile(gall, nlogical, gn, srcn, m0n, n0n);
      gemm4w_prologue(lds, srcn, wave, nt);
    }
  }

  epi_ops = gemm8p_store_tile(g, acc, m0 + 64 * wr, n0 + 64 * wc, lane, bias4);
  if (epi_ops < 0) {
#pragma unroll
    for (int ha = 0; ha < 2; ++ha) {
      typename AuxPre<bf16>::V upre[4][4];
      f32x4 blk[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        blk[i][0] = acc[ha][0][i][0]; blk[i][1] = acc[ha][0][i][1];
        blk[i][2] = acc[ha][1][i][0]; blk[i][3] = acc[ha][1][i][1];
      }
      gemm_epilogue<bf16, false, false>(g, blk, m0 + 128 * ha + 64 * wr, n0 + 64 * wc, 0, lane, bias4, nullptr, upre, false);
    }
  }
  if (g.dbg && tid == 0) {
    const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
    unsigned long long* d = g.dbg + (size_t)logical * 8;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = t_issued;
    d[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); d[5] = t_issued; d[6] = t_loop_end;
    d[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15;    // XCC id
  }
  if (!more) break;
  logical = nlogical;
  g = gall;
  gemm4w_tile(gall, logical, g, src, m0, n0);
  }                                          // ---- next tile
}

}  // namespace missm
