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

__global__ __launch_bounds__(256, 2) void gemm4w_kernel(GemmArgs gall) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 5 slots x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int nt = gall.K >> 6;                // K tiles (host guarantees K % 64 == 0, K >= 128)

  // ---- tile of this workgroup, its group's operands (grouped launch), buffer descriptors
  GemmArgs g = gall;                         // (scalar fields only are ever read through this copy)
  int tm, tn;
  tile_of(xcd_remap((int)blockIdx.x, ntiles), gall.tiles_m, gall.tiles_n, gall.group_m, tm, tn);
  if (gall.ngroups > 1) {
    const int gi = tm / gall.group_tiles_m;
    tm -= gi * gall.group_tiles_m;
    select_group(g, gall, gi);
  }
  const int m0 = tm * 256, n0 = tn * 128;
  Gemm4wSrc src;
  {
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
  const f32x4 bias4 = prefetch_bias(g, n0 + wc * 64, 0, lane);   // (older than every half-tile request)

  // ---- prologue: half tiles s = 0 .. 4  (B0 A0_0 A1_0 | B1 A0_1)
  stage4w<0>(lds, 0, src, wave, 0, true);
  stage4w<1>(lds, 1, src, wave, 0, true);
  stage4w<2>(lds, 2, src, wave, 0, true);
  stage4w<0>(lds, 3, src, wave, 128, nt > 1);
  stage4w<1>(lds, 4, src, wave, 128, nt > 1);

  // ---- fragment read addresses (bytes): row * 128 + ((4 ks + lg) ^ (row & 7)) * 16; + slot base and 16-row-tile immediates
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aoff[2], boff[2];                 // [k step]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned sw = (unsigned)(((ks * 4 + lg) ^ (li & 7)) << 4);
    aoff[ks] = lds0 + (unsigned)(wr * 64 + li) * 128u + sw;
    boff[ks] = lds0 + (unsigned)(wc * 32 + li) * 128u + sw;
  }

  unsigned long long t_start = 0, t_loop = 0, t_loop_end = 0;  // diagnostic runs only (tools/gemm_timeline.py)
  if (g.dbg) t_start = __builtin_amdgcn_s_memrealtime();
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
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // B0, A0_0, A1_0 have landed (this wave's pieces) ...
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

  if (gemm8p_store_tile(g, acc, m0 + 64 * wr, n0 + 64 * wc, lane, bias4) < 0) {
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
    unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
    d[0] = t_start; d[1] = t_loop; d[2] = t_loop_end; d[3] = t_issued;
    d[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); d[5] = t_issued; d[6] = t_loop_end;
    d[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15;    // XCC id
  }
}

// =====================================================================================================================
// CONTINUOUS form (round 3): 512 resident workgroups (two per CU) walk the tiles b, b + 512, ... and the K loop RUNS ON ACROSS TILE
// BOUNDARIES - the half tiles behind a tile's last K tile are the next tile's first ones, so the ring never drains, there is no
// prologue burst and no relaunch.  Why: in-kernel stamps of the one-tile form (tools/gemm_timeline.py) show a 12.7 us main loop and a
// 4.5 us epilogue per 256 x 128 tile with two tiles in flight per CU - 15 % less CU time per output than the 8-phase kernel, whose
// epilogue (store path: 34 - 52 GB/s per CU) is serial with its main loop - and then 3.5 - 5 us of EMPTY slot between two workgroups
// (a workgroup retires when its stores have drained, its successor starts cold): 121 us of work in a 157 us QKV launch.  A resident
// grid that requests the next tile's five prologue half tiles before the epilogue (MISSM_GEMM_PERSIST4W) was slower still: twenty
// requests per wave queued in front of the sixteen stores.  Here only the two requests of the next K tile's phase 0 stand in front of
// the stores (their slots are free: the fragments they held are in registers), and the epilogue's operations are counted into the two
// waits that follow it (loads and stores share vmcnt in issue order on gfx950: a wait for a load younger than the stores is a wait for
// the stores; the first such wait comes one and a half K tiles after the epilogue).
// Request schedule in global K-tile numbers g (phase 0 of g: A1(g + 1), B(g + 2); phase 1: A0(g + 2)); at a tile's first K tile g0
// phase 0 requests nothing - its two requests were issued in front of the previous epilogue:
//   ... phase 1 (g0 - 1): A0(g0 + 1) | barrier | A1(g0 + 1), B(g0 + 2) | EPILOGUE (E operations) | bias | phase 0 (g0): - | W0 | phase 1
//   (g0): A0(g0 + 2) | W1 | phase 0 (g0 + 1): A1(g0 + 2), B(g0 + 3) | W2 ...
//   W0 needs A0(g0 + 1): younger = 8 + E (+ bias) -> vmcnt(8 + E); W1 needs A1(g0 + 1): younger = 4 + E + 4 -> vmcnt(8 + E); W2 = vmcnt(8).
// The fragments of the next tile's first K tile (A0, B: read one phase ahead as always) stay in registers across the epilogue.
// Host guarantees: K % 64 == 0, K >= 192 (three K tiles: the requests two K tiles ahead never reach past the NEXT tile).
// =====================================================================================================================
__global__ __launch_bounds__(256, 2) void gemm4wc_kernel(GemmArgs gall) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 5 slots x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  const int ntiles = gall.tiles_m * gall.tiles_n;
  const int nt = gall.K >> 6;

  int bid = blockIdx.x;                      // hardware-order id of the tile in hand (logical id = xcd_remap(bid))
  GemmArgs g = gall;
  Gemm4wSrc src;
  int m0, n0;
  gemm4w_tile(gall, xcd_remap(bid, ntiles), g, src, m0, n0);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = 32 * wave + 8 * q + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    src.va[q] = (unsigned)r * (unsigned)gall.lda * 2u + (unsigned)c * 16u;
    const int hb = r >> 6, rp = r & 63;
    const int col = 64 * (rp >> 5) + 4 * (rp & 15) + 2 * hb + ((rp >> 4) & 1);
    src.vb[q] = (unsigned)col * (unsigned)gall.ldb * 2u + (unsigned)c * 16u;
  }
  gemm4w_prologue(lds, src, wave, nt);

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned sw = (unsigned)(((ks * 4 + lg) ^ (li & 7)) << 4);
    aoff[ks] = lds0 + (unsigned)(wr * 64 + li) * 128u + sw;
    boff[ks] = lds0 + (unsigned)(wc * 32 + li) * 128u + sw;
  }
  // the tile after this one (descriptors kept in scalar registers: the K loop requests from it behind its last two K tiles)
  int nbid = bid + (int)gridDim.x;
  bool more = nbid < ntiles;                 // wave-uniform
  GemmArgs gn = gall;
  Gemm4wSrc srcn = src;
  int m0n = 0, n0n = 0;
  if (more) gemm4w_tile(gall, xcd_remap(nbid, ntiles), gn, srcn, m0n, n0n);

  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // B0, A0_0, A1_0 of the first tile have landed (this wave's pieces) ...
  __builtin_amdgcn_s_barrier();                                // ... everybody's

  bf16x8 fa[2][4][2], fb[2][2][2];           // A fragments, double-buffered [buffer][i][ks]; B [hb][j][ks]
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
// wait for everything but the youngest 8 requests and the E operations of the epilogue that lies between them and the target
#define MISSM_4W_WAIT(E)                                                                      \
  if ((E) == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              \
  else if ((E) == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                       \
  else if ((E) == 32) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");                       \
  else if ((E) == 48) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");                       \
  else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");

  int sB = 0, sA0 = 1, sA1 = 2;              // slots of the K tile in hand: (3 g + h) mod 5, g counted over ALL tiles of this workgroup
  MISSM_4W_READ_B(sB)
  MISSM_4W_READ_A(sA0, 0)
  MISSM_4W_FENCE_AB(0)
  __builtin_amdgcn_s_barrier();              // B(0), A0(0) are in registers everywhere: their slots are free
  // phase 0 requests of the first K tile: A1(1) -> B(0)'s slot, B(2) -> A0(0)'s slot (K >= 192: both in this tile)
  stage4w<2>(lds, sB, src, wave, 128, true);
  stage4w<0>(lds, sA0, src, wave, 256, true);
  int epi = 0;                               // vector-memory operations of the epilogue in front of this tile's first K tile

  for (;;) {                                 // ---- one output tile per iteration
    unsigned long long t_start = 0, t_loop_end = 0;
    if (g.dbg) t_start = __builtin_amdgcn_s_memrealtime();
    f32x4 bias4 = prefetch_bias(g, n0 + wc * 64, 0, lane);
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nt; ++t) {
      // K tiles t + 1 and t + 2 of this tile - or the first ones of the next tile
      const bool in1 = t + 1 < nt, in2 = t + 2 < nt;
      const int kb1 = (in1 ? t + 1 : t + 1 - nt) * 128, kb2 = (in2 ? t + 2 : t + 2 - nt) * 128;
      const bool live1 = in1 || more, live2 = in2 || more;
      const __amdgpu_buffer_rsrc_t rA1_1 = live1 ? (in1 ? src.a[1] : srcn.a[1]) : src.none;     // A1(t + 1)
      const __amdgpu_buffer_rsrc_t rB_2 = live2 ? (in2 ? src.b : srcn.b) : src.none;           // B(t + 2)
      const __amdgpu_buffer_rsrc_t rA0_2 = live2 ? (in2 ? src.a[0] : srcn.a[0]) : src.none;     // A0(t + 2)
      using lptr = __attribute__((address_space(3))) void*;
      const int nB = sB + 3 >= 5 ? sB - 2 : sB + 3, nA0 = sA0 + 3 >= 5 ? sA0 - 2 : sA0 + 3, nA1 = sA1 + 3 >= 5 ? sA1 - 2 : sA1 + 3;
      char* const dB = lds + sB * 16384 + wave * 4096;
      char* const dA0 = lds + sA0 * 16384 + wave * 4096;
      char* const dA1 = lds + sA1 * 16384 + wave * 4096;
      const bool first = t == 0;             // phase 0 of a tile's first K tile requests nothing (issued in front of the epilogue)
      // ---- phase 0: A half 0 x B.  Side: requests A1(t + 1) -> B(t)'s slot, B(t + 2) -> A0(t)'s slot; reads of A1(t)
      __builtin_amdgcn_s_setprio(1);
      static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        MISSM_4W_MF(0, 0, n)
        if constexpr (n % 2 == 1) {
          constexpr int k = n / 2;
          if constexpr (k % 2 == 0) {
            if (!first) {
              if constexpr (k / 2 < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1_1, (lptr)(dB + (k / 2) * 1024), 16, src.va[k / 2], kb1, 0, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB_2, (lptr)(dA0 + (k / 2 - 4) * 1024), 16, src.vb[k / 2 - 4], kb2, 0, 0);
            }
          } else {
            MISSM_4W_RA(sA1, 1, k / 2)
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_s_setprio(0);
      MISSM_4W_FENCE_A(1)
      if (first) { MISSM_4W_WAIT(epi) } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- phase 1: A half 1 x B.  Side: request A0(t + 2) -> A1(t)'s slot, reads of A0(t + 1), then of B(t + 1)
      __builtin_amdgcn_s_setprio(1);
      static_for<0, 32>([&](auto n_) {
        constexpr int n = decltype(n_)::value;
        MISSM_4W_MF(1, 1, n)
        if constexpr (n % 2 == 1) {
          constexpr int k = n / 2;
          if constexpr (k < 8) {
            if constexpr (k % 2 == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0_2, (lptr)(dA1 + (k / 2) * 1024), 16, src.va[k / 2], kb2, 0, 0);
            else MISSM_4W_RA(nA0, 0, k / 2)
          } else {
            if constexpr (k % 2 == 0) MISSM_4W_RA(nA0, 0, 4 + (k - 8) / 2)
            else MISSM_4W_RB(nB, (k - 8) / 2)
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      __builtin_amdgcn_s_setprio(0);
      MISSM_4W_RB(nB, 4) MISSM_4W_RB(nB, 5) MISSM_4W_RB(nB, 6) MISSM_4W_RB(nB, 7)
      MISSM_4W_FENCE_AB(0)
      if (first) { MISSM_4W_WAIT(epi) } else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      sB = nB; sA0 = nA0; sA1 = nA1;
    }
    if (g.dbg) t_loop_end = __builtin_amdgcn_s_memrealtime();
    asm volatile("" : "+v"(bias4));            // (the compiler's wait for the bias vector lands here, not behind the requests below)
    // The next tile's A0(0) / B(0) fragments are in registers, their slots free: its phase 0 requests go out in front of the stores.
    if (more) {
      stage4w<2>(lds, sB, srcn, wave, 128, true);
      stage4w<0>(lds, sA0, srcn, wave, 256, true);
    }
    epi = gemm8p_store_tile(g, acc, m0 + 64 * wr, n0 + 64 * wc, lane, bias4);
    if (epi < 0) {
      epi = 0;                               // (guarded path: an unknown number of operations - wait for all of them)
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
      unsigned long long* d = g.dbg + (size_t)xcd_remap(bid, ntiles) * 8;
      d[0] = t_start; d[1] = t_start; d[2] = t_loop_end; d[3] = t_issued;
      d[4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); d[5] = t_issued; d[6] = t_loop_end;
      d[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 15;
    }
    if (!more) break;
    bid = nbid;
    g.A = gn.A; g.B = gn.B; g.C = gn.C; g.bias = gn.bias; g.resid = gn.resid; g.aux_in = gn.aux_in; g.aux_out = gn.aux_out;
    src.a[0] = srcn.a[0]; src.a[1] = srcn.a[1]; src.b = srcn.b;
    m0 = m0n; n0 = n0n;
    nbid = bid + (int)gridDim.x;
    more = nbid < ntiles;
    if (more) { gn = gall; gemm4w_tile(gall, xcd_remap(nbid, ntiles), gn, srcn, m0n, n0n); }
  }
#undef MISSM_4W_MF
#undef MISSM_4W_RA
#undef MISSM_4W_RB
#undef MISSM_4W_READ_A
#undef MISSM_4W_READ_B
#undef MISSM_4W_FENCE_A
#undef MISSM_4W_FENCE_AB
#undef MISSM_4W_WAIT
}

}  // namespace missm
