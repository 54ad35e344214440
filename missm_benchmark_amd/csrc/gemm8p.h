// 256x256 NT GEMM tile as an 8-wave, 8-phase software pipeline (bf16 MFMA 16x16x32) - included by gemm.hip.
//
// Why a second tile kernel: the 16-wave 256x256 kernel in gemm.hip drains its LDS-DMA (s_waitcnt vmcnt(0)) and meets at a
// workgroup barrier once per K tile with every wave in the same phase (54-57 % MFMA-busy main loop).  Here
//   * a wave owns 128 x 64 outputs (128 accumulator VGPRs): half the LDS -> register bytes per flop of a 64 x 64 wave block;
//   * operands travel as eight 16 KiB HALF tiles per two K tiles (A rows 0-127 / 128-255 and two interleaved halves of the B
//     rows, K tile 64): one half tile is requested per phase, SEVEN half tiles ahead of its use, straight into LDS
//     (buffer_load ... lds), and is waited for with a COUNTED s_waitcnt vmcnt(6) once per K tile - never 0 inside the loop;
//   * a phase = {fragment reads of the half tile that became new, one half-tile request} | barrier | 16 MFMAs (one 64 x 32
//     quadrant x K 64) | barrier; waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave's MFMAs cover the
//     other's LDS reads and DMA issue (/opt/skills/guides/cdna_hip_programming.md section 5, "8-phase template").
// Hazards (derivation in DESIGN.md section 4):
//   RAW  a half tile is read at the earliest one phase after the vmcnt wait that retires it (the wait sits before the
//        phase's first barrier, every wave passes it before any wave of either group reads);
//   WAR  half tile s + 8 reuses the slot of s and is requested in phase s + 1: two or more phases after the last read of s for
//        three of the four half tiles; for the first one of a K tile (B half 0, read in the phase before) its four reads are
//        retired by a counted lgkmcnt BEFORE the reading phase's first barrier.
// Fragment reads are inline assembly: hipcc otherwise guards every LDS read behind an in-flight LDS-DMA with vmcnt(0).
#pragma once

namespace missm {

template <int OFF> __device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

struct Gemm8pSrc {
  __amdgpu_buffer_rsrc_t a[2], b[2], none;   // A rows of half 0 / 1, B rows of half 0 / 1, zero-record descriptor (drops the load)
  unsigned va[2], vb[2];                     // per-lane byte offsets of this wave's two 1 KiB pieces inside a half tile
};

// one half tile = 16 pieces of 1 KiB; wave w moves pieces 2w, 2w + 1 (LDS rows 8 * piece .. + 7, 128 bytes of k each)
template <int SLOT, int H>
__device__ __forceinline__ void stage_half(char* lds, const Gemm8pSrc& s, int wave, int kbyte, bool live) {
  using lptr = __attribute__((address_space(3))) void*;
  constexpr bool IS_A = (H == 1 || H == 3);
  constexpr int HALF = H >> 1;
  const __amdgpu_buffer_rsrc_t r = live ? (IS_A ? s.a[HALF] : s.b[HALF]) : s.none;
  char* dst = lds + SLOT * 16384 + wave * 2048;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)dst, 16, IS_A ? s.va[0] : s.vb[0], kbyte, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr)(dst + 1024), 16, IS_A ? s.va[1] : s.vb[1], kbyte, 0, 0);
}

// STAGGER: waves 4-7 run one barrier behind waves 0-3
template <bool STAGGER>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char lds[];   // 8 slots x 16 KiB; slot = (4 * (K tile & 1) + h), h: B0 A0 B1 A1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int li = lane & 15, lg = lane >> 4;

  const int ntiles = g.tiles_m * g.tiles_n;
  int tm, tn;
  tile_of(xcd_remap(blockIdx.x, ntiles), g.tiles_m, g.tiles_n, g.group_m, tm, tn);
  const int m0 = tm * 256, n0 = tn * 256;
  const int nt = g.K >> 6;                   // K tiles (host guarantees K % 128 == 0: whole pairs)

  // ---- global -> LDS addressing.  LDS row r of a half tile holds 128 bytes of k, chunk c stored at c ^ (r & 7).
  //  A half ha : LDS row r <-> C row   m0 + 128 ha + r                       (wave wr reads rows 64 wr + 16 i + li)
  //  B half hb : LDS row r <-> C column n0 + 64 (r >> 5) + 4 (r & 15) + 2 hb + ((r >> 4) & 1)
  //              (wave wc reads rows 32 wc + 16 j + li: with both halves a lane owns 4 CONSECUTIVE columns 64 wc + 4 li + 0..3)
  Gemm8pSrc src;
  {
    const bf16* A = static_cast<const bf16*>(g.A);
    const bf16* B = static_cast<const bf16*>(g.B);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ra = g.M - (m0 + 128 * h), rb = g.N - (n0 + 2 * h);
      const unsigned na = ra <= 0 ? 0u : (unsigned)min(ra, 128) * (unsigned)g.lda * 2u;
      const unsigned nb = rb <= 0 ? 0u : (unsigned)min(rb, 254) * (unsigned)g.ldb * 2u;
      src.a[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A + (size_t)(m0 + 128 * h) * g.lda), 0, na, 0x00020000);
      src.b[h] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(B + (size_t)(n0 + 2 * h) * g.ldb), 0, nb, 0x00020000);
    }
    src.none = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(A), 0, 0, 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = 16 * wave + 8 * q + (lane >> 3);           // LDS row inside the half tile
      const int c = (lane & 7) ^ (r & 7);                      // the k chunk that lands at stored position lane & 7
      src.va[q] = (unsigned)r * (unsigned)g.lda * 2u + (unsigned)c * 16u;
      const int col = 64 * (r >> 5) + 4 * (r & 15) + ((r >> 4) & 1);
      src.vb[q] = (unsigned)col * (unsigned)g.ldb * 2u + (unsigned)c * 16u;
    }
  }

  // ---- fragment read addresses (bytes): row * 128 + ((4 ks + lg) ^ (row & 7)) * 16; + slot and 16-row-tile immediates
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  unsigned aaddr[2][2], baddr[2][2];         // [K tile parity][k step]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const unsigned sw = (unsigned)(((ks * 4 + lg) ^ (li & 7)) << 4);
    aaddr[0][ks] = lds0 + (unsigned)(wr * 64 + li) * 128u + sw;
    baddr[0][ks] = lds0 + (unsigned)(wc * 32 + li) * 128u + sw;
    aaddr[1][ks] = aaddr[0][ks] + 65536u;
    baddr[1][ks] = baddr[0][ks] + 65536u;
  }

  f32x4 acc[2][2][4][2];                     // [A half][B half][16-row tile][16-column tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4 bias4 = prefetch_bias(g, n0 + wc * 64, 0, lane);

  // ---- prologue: half tiles 0 .. 6 (K tile 0 and three quarters of K tile 1)
  stage_half<0, 0>(lds, src, wave, 0, true);
  stage_half<1, 1>(lds, src, wave, 0, true);
  stage_half<2, 2>(lds, src, wave, 0, true);
  stage_half<3, 3>(lds, src, wave, 0, true);
  stage_half<4, 0>(lds, src, wave, 128, nt > 1);
  stage_half<5, 1>(lds, src, wave, 128, nt > 1);
  stage_half<6, 2>(lds, src, wave, 128, nt > 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");            // K tile 0 has landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();                                // ... everybody's
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();        // waves 4-7 fall one barrier behind (re-joined after the loop)

  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];     // A half in use [i][ks]; B half 0 / 1 [j][ks]

#define MISSM_8P_MFMA(HA, HB, FB)                                                             \
  __builtin_amdgcn_s_setprio(1);                                                              \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                           \
        acc[HA][HB][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], FB[j][ks], acc[HA][HB][i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0);

#define MISSM_8P_READ_A(PAR, SLOT_H)                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    fa[0][ks] = lds_read128<(SLOT_H) * 16384 + 0 * 2048>(aaddr[PAR][ks]);                    \
    fa[1][ks] = lds_read128<(SLOT_H) * 16384 + 1 * 2048>(aaddr[PAR][ks]);                    \
    fa[2][ks] = lds_read128<(SLOT_H) * 16384 + 2 * 2048>(aaddr[PAR][ks]);                    \
    fa[3][ks] = lds_read128<(SLOT_H) * 16384 + 3 * 2048>(aaddr[PAR][ks]);                    \
  }
#define MISSM_8P_READ_B(PAR, SLOT_H, FB)                                                      \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                          \
    FB[0][ks] = lds_read128<(SLOT_H) * 16384 + 0 * 2048>(baddr[PAR][ks]);                    \
    FB[1][ks] = lds_read128<(SLOT_H) * 16384 + 1 * 2048>(baddr[PAR][ks]);                    \
  }
#define MISSM_8P_FENCE_ALL()                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                         \
               : "+v"(fa[0][0]), "+v"(fa[1][0]), "+v"(fa[2][0]), "+v"(fa[3][0]), "+v"(fa[0][1]), "+v"(fa[1][1]), "+v"(fa[2][1]), \
                 "+v"(fa[3][1]), "+v"(fb0[0][0]), "+v"(fb0[1][0]), "+v"(fb0[0][1]), "+v"(fb0[1][1]), "+v"(fb1[0][0]),            \
                 "+v"(fb1[1][0]), "+v"(fb1[0][1]), "+v"(fb1[1][1]));                          \
  __builtin_amdgcn_sched_barrier(0);

  // One K tile = 4 phases.  PAR = parity of the K tile (its slots are 4 PAR + h).  Phase p requests half tile 4 t + p + 7:
  //   p = 0 -> (t + 1, A1) slot 4 (1 - PAR) + 3 ; p = 1, 2, 3 -> (t + 2, B0 / A0 / B1) slots 4 PAR + 0 / 1 / 2
  auto ktile = [&](auto par_tag, int t) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int S = 4 * PAR;               // first slot of this K tile (in units of 16 KiB relative to the parity base)
    const int kb1 = (t + 1) * 128, kb2 = (t + 2) * 128;
    const bool live1 = t + 1 < nt, live2 = t + 2 < nt;
    // ---- phase 0: A half 0 x B half 0
    MISSM_8P_READ_B(PAR, 0, fb0)
    MISSM_8P_READ_A(PAR, 1)
    stage_half<4 * (1 - PAR) + 3, 3>(lds, src, wave, kb1, live1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");         // the four B-half-0 reads are back: its slot is re-filled next phase
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 1: A half 0 x B half 1
    MISSM_8P_READ_B(PAR, 2, fb1)
    stage_half<S + 0, 0>(lds, src, wave, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(0, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: A half 1 x B half 1
    MISSM_8P_READ_A(PAR, 3)
    stage_half<S + 1, 1>(lds, src, wave, kb2, live2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_FENCE_ALL()
    MISSM_8P_MFMA(1, 1, fb1)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3: A half 1 x B half 0 (fragments kept since phase 0); the next K tile must have landed before its reads
    stage_half<S + 2, 2>(lds, src, wave, kb2, live2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    MISSM_8P_MFMA(1, 0, fb0)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  // note on slot arithmetic: READ/stage offsets above are relative to the parity base (aaddr[PAR] carries + 64 KiB for PAR = 1),
  // stage_half takes ABSOLUTE slots.
  for (int t = 0; t < nt; t += 2) {
    ktile(std::integral_constant<int, 0>{}, t);
    ktile(std::integral_constant<int, 1>{}, t + 1);
  }
#undef MISSM_8P_MFMA
#undef MISSM_8P_READ_A
#undef MISSM_8P_READ_B
#undef MISSM_8P_FENCE_ALL
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();        // waves 0-3 wait for the lagging group
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the (dropped) requests past the last K tile

  // ---- epilogue: two 64 x 64 blocks per wave (A half 0 / 1), a lane owns rows 4 lg + r of each 16-row tile and the four
  // consecutive columns 64 wc + 4 li + {0, 1 (B half 0), 2, 3 (B half 1)}: the same register picture as gemm_kernel's
  typename AuxPre<bf16>::V upre[4][4];
#pragma unroll
  for (int ha = 0; ha < 2; ++ha) {
    f32x4 blk[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      blk[i][0] = acc[ha][0][i][0]; blk[i][1] = acc[ha][0][i][1];
      blk[i][2] = acc[ha][1][i][0]; blk[i][3] = acc[ha][1][i][1];
    }
    gemm_epilogue<bf16, false, false>(g, blk, m0 + 128 * ha + 64 * wr, n0 + 64 * wc, 0, lane, bias4, nullptr, upre, false);
  }
}

}  // namespace missm
