// fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), the engine under every O(n^3)
// step of the GP path: the SYRK/GEMM trailing updates of the Cholesky (regression.py:36,49),
// the TRSM panels (regression.py:38), V^T V (regression.py:41), trtri/lauum (regression.py:60).
//
//   D = alpha * op(A) op(B) + beta * Cin (+ diag_add on the diagonal),  row-major C/D.
//
// Structure (MI355X-first; measured: the f64 MFMA issues once per 64 cycles per SIMD, 77.7 TFLOP/s
// chip-wide, and only with VGPR accumulators -- AGPR accumulators run at 138 cycles):
//   * 256-thread workgroup = 2 x 2 waves, each wave TM x TN MFMA tiles of 16 x 16
//     (TM = TN = 4: 128 x 128 block tile, 128 accumulator VGPRs; TM = TN = 2: 64 x 64 block tile
//     for launches too small to fill 256 CUs), BK = 16.
//   * interior k-tiles are brought in by LDS-DMA (global_load_lds, 16 B per lane, no staging
//     registers, no ds_write) into a double-buffered LDS image -> one barrier per k-tile, the DMA
//     of tile t+1 flies under the MFMAs of tile t.  The image is unpadded (DMA writes lane-linear)
//     and XOR-swizzled through the per-lane SOURCE address so the ds_read_b64 fragment reads are
//     bank-conflict free (bank = (addr/4) % 64 per 32-lane half, MI355X_MICROARCH.md section LDS):
//       KC operands (k contiguous in HBM): image [rows][8 slots of 16 B], slot ^= (row >> 1) & 7
//       RC operands (row index contiguous) : image [16 k][rows/2 slots],   slot ^= (k & 1) << 3
//   * edge tiles and tiles crossed by the diagonal of a triangular operand use guarded register
//     loads + ds_write_b128 into the same image.
//   * fragments for MFMA step kk+1 are read from LDS before the MFMAs of step kk are issued.
//   * 1-D grid, remapped so that each XCD (blocks id % 8) walks 8 x 8 super-tiles: the 64 blocks
//     resident on one XCD share 8 A panels and 8 B panels in its private L2.
//
// Triangular operands (a_tri / b_tri: the STORED matrix is lower triangular and starts on its
// diagonal) restrict the k range per tile and zero-mask inside the boundary tile, so blocks that
// lie strictly above the diagonal are never read (they may hold anything, including NaN).
#include <cstdlib>

#include "gpx_internal.h"


namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BK = 16, NTHREADS = 256;
constexpr int SYNC_KT = 128;            // default k-tiles between two pacing points of a super-tile (a power of two)
constexpr int SYNC_MAX_POLLS = 20000;   // x ~0.2 us: a workgroup gives up waiting after ~4 ms and goes on alone

// LDS-DMA of 16 bytes per lane: global `src` (per lane) -> LDS `dst_wave` + lane * 16 (wave-uniform base).
__device__ __forceinline__ void glds16(const char* src, double* dst_wave) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave, 16, 0, 0);
}

// The same DMA in its buffer form: wave-uniform base in a buffer resource (4 SGPRs), per-lane 32-bit byte
// offset in ONE loop-invariant VGPR.  The k advance goes into the resource's base with scalar adds, so a
// DMA piece costs no VALU instruction at all (with global_load_lds the per-lane offset had to be copied
// through a v_mov per piece to keep the compiler from rebuilding 64-bit per-lane pointers: 8 VALU
// instructions per k-tile taken out of the MFMA issue stream).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const char* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0xffffffff, 0x00020000);
}

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, double* dst_wave) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst_wave, 16, voffset, 0, 0, 0);
}

// One DMA piece of an operand tile.  K-contiguous operands take the buffer form; operands whose k index
// is the strided one keep the global form (measured: the buffer form is 1.2 % faster on potrf's and
// trtri's launches and 1.3 % slower on lauum's, whose two operands are both of the second kind).
template <int LAY>
__device__ __forceinline__ void dma16(const char* base, __amdgpu_buffer_rsrc_t rsrc, unsigned off, double* dst_wave) {
  if constexpr (LAY == GPX_LAY_KC) {
    blds16(rsrc, off, dst_wave);
  } else {
    asm volatile("" : "+v"(off));  // keep the offset 32-bit: saddr + voffset addressing, no 64-bit VALU add
    glds16(base + off, dst_wave);
  }
}

// Tell the compiler a pointer is wave-uniform (kept in SGPRs), so that `p + lane_offset` becomes the
// saddr + 32-bit voffset form of the load instead of 64-bit VALU adds.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// One operand's (R rows) x (BK) k-tile as an LDS image of 16-byte slots.
template <int LAY, int R>
struct Tile {
  static constexpr int ELEMS = R * BK;
  static constexpr int SLOTS = ELEMS / 2;
  static constexpr int PER_THREAD = SLOTS / NTHREADS;  // 4 (R = 128) or 2 (R = 64)
  static constexpr int SPR = R / 2;                    // slots per k row (RC)

  // image slot p -> coordinates (op-row r, k) of its first double
  static __device__ __forceinline__ void decode(int p, int& r, int& k) {
    if (LAY == GPX_LAY_KC) {
      const int row = p >> 3;
      r = row;
      k = 2 * ((p & 7) ^ ((row >> 1) & 7));
    } else {
      const int kr = p / SPR;
      k = kr;
      r = 2 * ((p % SPR) ^ ((kr & 1) << 3));
    }
  }

  // element (r, k) -> index (doubles) in the image
  static __device__ __forceinline__ int index(int r, int k) {
    if (LAY == GPX_LAY_KC) return r * BK + ((((k >> 1) ^ ((r >> 1) & 7))) << 1) + (k & 1);
    return k * R + ((((r >> 1) ^ ((k & 1) << 3))) << 1) + (r & 1);
  }

  // Fragment addressing split into a per-lane base (computed once) and compile-time offsets.
  // For tile T (16 op-rows) and MFMA k-step kk the element read by lane (i16, g4) is
  // index(T*16 + i16, 4*kk + g4) == frag_base(frag_sel(T, kk), i16, g4) + frag_off(T, kk).
  static constexpr int NBASE = (LAY == GPX_LAY_KC) ? BK / 4 : 2;
  static __device__ __forceinline__ int frag_base(int sel, int i16, int g4) {
    if (LAY == GPX_LAY_KC) {  // sel = kk
      const int c = (g4 >> 1) ^ ((i16 >> 1) & 7);
      return i16 * BK + (((2 * sel) ^ c) << 1) + (g4 & 1);
    }
    const int b = g4 & 1;     // sel = T & 1: T ^ b == T + b (T even), T - b (T odd)
    return g4 * R + i16 + (sel ? -16 * b : 16 * b);
  }
  static constexpr __device__ __forceinline__ int frag_sel(int T, int kk) { return LAY == GPX_LAY_KC ? kk : (T & 1); }
  static constexpr __device__ __forceinline__ int frag_off(int T, int kk) {
    return LAY == GPX_LAY_KC ? T * 16 * BK : kk * 4 * R + 16 * T;
  }

  // Does the k-tile [k0, k0+BK) of the op-row tile [r0, r0+R) need the guarded loader?
  static __device__ __forceinline__ bool needs_guard(int r0, int Rext, int k0, int K, int tri) {
    if (r0 + R > Rext || k0 + BK > K) return true;
    if (!tri) return false;
    if (LAY == GPX_LAY_KC) return k0 + BK - 1 > r0;  // valid k <= r for every r >= r0
    return r0 + R - 1 > k0;                          // valid r <= k for every k >= k0
  }

  // boundary tile: guarded loads (zero fill) through registers
  static __device__ __forceinline__ void guarded(const double* __restrict__ P, long ld, int r0, int Rext,
                                                 int k0, int K, int tri, double* S, int t) {
    v2d v[PER_THREAD];
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) {
      int rr, kk;
      decode(q * NTHREADS + t, rr, kk);
      const int r = r0 + rr, k = k0 + kk;
      bool v0, v1;
      const double* src;
      if (LAY == GPX_LAY_KC) {  // pair along k; stored (row r, col k): valid k <= r
        const bool rin = r < Rext;
        v0 = rin && (k < K) && (!tri || k <= r);
        v1 = rin && (k + 1 < K) && (!tri || k + 1 <= r);
        src = P + (long)r * ld + k;
      } else {                  // pair along r; stored (row k, col r): valid r <= k
        const bool kin = k < K;
        v0 = kin && (r < Rext) && (!tri || r <= k);
        v1 = kin && (r + 1 < Rext) && (!tri || r + 1 <= k);
        src = P + (long)k * ld + r;
      }
      v2d x = {0.0, 0.0};
      if (v1) {  // v1 implies v0 (valid sets are prefixes along the contiguous index)
        x = *reinterpret_cast<const v2d*>(src);
      } else if (v0) {
        x[0] = src[0];
      }
      v[q] = x;
    }
#pragma unroll
    for (int q = 0; q < PER_THREAD; ++q) *reinterpret_cast<v2d*>(S + 2 * (q * NTHREADS + t)) = v[q];
  }

};

// Logical tile coordinates of this workgroup.  Blocks id % 8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH.md): the idx-th block of an XCD takes slot idx % 64 of super-tile
// (idx / 64) * 8 + xcd, so the 64 workgroups resident on one XCD work on one 8 x 8 super-tile
// (8 A panels + 8 B panels in its private L2) and super-tiles are dealt round-robin over the XCDs
// (balanced also when the work per tile varies, as with triangular operands).
// Traversal order of the super-tiles of a full (non-lower) launch.  With a triangular operand the
// work of a tile depends on its row or column; the heavy ones go first so that the last round of the
// launch holds only light tiles (longest-processing-time-first), and at any moment all XCDs work on
// tiles of similar weight.
enum { ORD_ROWS_SKEWED = 0, ORD_COLS_ASC = 1, ORD_COLS_DESC = 2, ORD_ROWS_DESC = 3, ORD_ROWS_ASC_TRI = 4 };

__host__ __device__ __forceinline__ void super_coords(unsigned st, int tiles_m, int tiles_n, int order, unsigned& sr,
                                                      unsigned& sc) {
  const unsigned stc = (tiles_n + 7) / 8, str_ = (tiles_m + 7) / 8;
  switch (order) {
    case ORD_COLS_ASC: sc = st / str_; sr = st % str_; break;
    case ORD_COLS_DESC: sc = stc - 1 - st / str_; sr = st % str_; break;
    case ORD_ROWS_DESC: sr = str_ - 1 - st / stc; sc = st % stc; break;
    default:
      // skewed columns: with stc a multiple of 8, plain st % stc would hand XCD x the same super-tile
      // columns on every row; the skew walks every XCD through all columns.
      sr = st / stc;
      sc = (st % stc + sr) % stc;
  }
}

// Logical tile coordinates of this workgroup.  Blocks id % 8 share an XCD (round-robin dispatch,
// MI355X_MICROARCH.md): the idx-th block of an XCD takes slot idx % 64 of super-tile
// (idx / 64) * 8 + xcd, so the 64 workgroups resident on one XCD work on one 8 x 8 super-tile
// (8 A panels + 8 B panels in its private L2) and super-tiles are dealt round-robin over the XCDs
// (balanced also when the work per tile varies, as with triangular operands).
__device__ __forceinline__ bool block_coords(int tiles_m, int tiles_n, int lower, unsigned nsuper, unsigned bid,
                                             int order, int& bi, int& bj) {
  if (nsuper == 0) {  // small launch: one block per tile, linear (row-major / packed lower) order
    if (lower) {
      unsigned r = (unsigned)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= bid) ++r;
      while (r * (r + 1) / 2 > bid) --r;
      bi = (int)r;
      bj = (int)(bid - r * (r + 1) / 2);
    } else {
      // One round of the chip (257..512 tiles, two workgroups per CU) with a triangular operand: the work of a
      // tile grows linearly along the rows (or columns), and ids c and 256 + c share a CU.  The second 256
      // ids walk the tiles backwards from the last one, so that every CU gets a light and a heavy tile
      // (row-major ids pair row r with row r + 8 of 16: the slowest CU holds 24/17 of the mean -- a 2048 x 4096
      // product with a triangular 2048-operand took 373 us against 245 us balanced).
      // (the remap itself is applied by the kernel before this call: it also serves packed lower launches)
      bi = (int)(bid / (unsigned)tiles_n);
      bj = (int)(bid % (unsigned)tiles_n);
    }
    return bi < tiles_m && bj < tiles_n;
  }
  const unsigned xcd = bid & 7, idx = bid >> 3;
  const unsigned st = (idx >> 6) * 8 + xcd, within = idx & 63;
  if (st >= nsuper) return false;
  unsigned sr, sc;
  if (lower) {
    sr = (unsigned)((sqrt(8.0 * (double)st + 1.0) - 1.0) * 0.5);
    while ((sr + 1) * (sr + 2) / 2 <= st) ++sr;
    while (sr * (sr + 1) / 2 > st) --sr;
    sc = st - sr * (sr + 1) / 2;
  } else {
    super_coords(st, tiles_m, tiles_n, order, sr, sc);
  }
  bi = (int)(sr * 8 + (within >> 3));
  bj = (int)(sc * 8 + (within & 7));
  if (bi >= tiles_m || bj >= tiles_n) return false;
  if (lower && bj > bi) return false;
  return true;
}

template <int ALAY, int BLAY, int TM, int TN>
__global__ __launch_bounds__(NTHREADS, (TM * TN >= 16 ? 2 : (TM * TN >= 8 ? 3 : 4))) void gemm_f64_kernel(const GemmDesc gin, int tiles_m, int tiles_n,
                                                                 unsigned nsuper, unsigned bid0, int order) {
  GemmDesc g = gin;
  if (gin.batch > 1) {   // uniform: this workgroup's product of the batch
    const long z = blockIdx.y;
    long oa = z * gin.stride_a, ob = z * gin.stride_b, od = z * gin.stride_d;
    if (gin.batch_inner > 0) {   // two-level batch, see GemmDesc
      const long zi = z % gin.batch_inner, zo = z / gin.batch_inner;
      oa = zi * gin.stride_a + zo * gin.stride2_a;
      ob = zi * gin.stride_b + zo * gin.stride2_b;
      od = zi * gin.stride_d + zo * gin.stride2_d;
    }
    g.A += oa;
    g.B += ob;
    g.D += od;
    if (g.Cin != nullptr) g.Cin += od;
  }
  constexpr int BM = 32 * TM, BN = 32 * TN;
  typedef Tile<ALAY, BM> TA;
  typedef Tile<BLAY, BN> TB;
  __shared__ __attribute__((aligned(16))) double smem[2 * (TA::ELEMS + TB::ELEMS)];

  int bi, bj;
  unsigned bid = blockIdx.x + bid0;
  int part = 0;
  if (g.split_s > 1) {   // split-K tail: split_s consecutive workgroups share one tile
    part = (int)(bid % (unsigned)g.split_s);
    bid = g.split_tile0 + bid / (unsigned)g.split_s;
  }
  // light / heavy pairing (see block_coords): this launch is one round of the chip, one block per tile, and the
  // tiles' work falls or rises monotonically with the id (a triangular operand)
  if (nsuper == 0 && order != ORD_ROWS_SKEWED && g.split_s <= 1 && bid0 == 0 && gridDim.x > 256 && gridDim.x <= 512 &&
      bid >= 256)
    bid = gridDim.x - 1 - (bid - 256);
  if (!block_coords(tiles_m, tiles_n, g.c_lower, nsuper, bid, order, bi, bj)) return;
  // Pacing (plain operands, super-tile mode, long K).  The 64 workgroups of a super-tile share 8 + 8 operand
  // panels through their XCD's 4 MiB L2, which holds a window of only 16 k-tiles of those panels; nothing
  // keeps the workgroups in step, and after the first round of a launch they start whenever a slot frees,
  // so partners are rarely inside one window (measured: L2 hit rate 36 % on M = N = K = 16384, where lockstep
  // would give 87 %).  Every SYNC_KT k-tiles thread 0 checks in at the super-tile's counter and waits --
  // bounded, so a missing partner can only cost time -- until all live workgroups of the super-tile have
  // reached the same k.  Workgroups are dispatched in id order, so all members of a super-tile are resident
  // (or about to be) whenever one of them is.
  unsigned* pace = nullptr;
  unsigned pace_live = 0;
  if (g.sync_cnt != nullptr && nsuper != 0) {
    const unsigned xcd = bid & 7, idx = bid >> 3;
    const unsigned st = (idx >> 6) * 8 + xcd;
    int li, lj;
    const bool alive = block_coords(tiles_m, tiles_n, g.c_lower, nsuper, (((idx >> 6) << 6) + (threadIdx.x & 63)) * 8 + xcd,
                                    order, li, lj);
    pace_live = (unsigned)__popcll(__ballot(alive));   // (every wave computes it; thread 0 uses it)
    pace = g.sync_cnt + st;
  }
  if (g.cb_off != nullptr) {   // column-block output: this tile's slab; tiles above the slab's first row do not exist
    const int c = (bj * BN) / g.cb_nb;
    // rows above the slab's first row do not exist; inside the slab's first cb_nb rows (its DIAGONAL tile) the
    // tiles strictly above that tile's diagonal are never read by the factorisation and are skipped as well, so
    // the launch does exactly the lower trapezoid's flops (full squares cost 4.7 % more at cb_nb = 2048)
    // (bit 0 of the table entry flags a slab whose first tile IS the diagonal tile: on a P x Q grid the slab of a
    // column this process row does not own the diagonal tile of starts with a full tile below the diagonal)
    const int r0e = g.cb_row0[c];
    const int rel_last = g.cb_rowbase + bi * BM + BM - 1 - (r0e & ~1);   // last row of the tile, slab-relative
    // (workgroup-uniform, before any barrier; the skip works on 128 x 128 blocks whatever the tile size)
    if (rel_last < 0 || ((r0e & 1) && (rel_last | 127) < ((bj * BN - c * g.cb_nb) & ~127))) return;
    g.D += g.cb_off[c] + (long)g.cb_rowbase * g.cb_nb - (long)c * g.cb_nb;
    g.Cin = g.D;
  }
  if (g.c_upper) {   // mirror image of the lower enumeration (symmetric product): tile (bj, bi) on or above the diagonal
    const int tmp = bi;
    bi = bj;
    bj = tmp;
  }
  const int i0 = bi * BM, j0 = bj * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int i16 = lane & 15, g4 = lane >> 4;

  int kb = 0, ke = g.K;
  if (g.a_tri) {
    if (ALAY == GPX_LAY_KC) ke = min(ke, i0 + BM); else kb = max(kb, i0);
  }
  if (g.b_tri) {
    if (BLAY == GPX_LAY_KC) ke = min(ke, j0 + BN); else kb = max(kb, j0);
  }
  kb &= ~(BK - 1);
  if (g.split_s > 1) {   // this workgroup's slice of K (a multiple of BK; plain operands only)
    const int ks = ((g.K + g.split_s * BK - 1) / (g.split_s * BK)) * BK;
    kb = part * ks;
    ke = min(g.K, kb + ks);
  }

  v4d acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  if (kb < ke) {
    // Everything the k loop needs per lane is computed here, once: the f64 MFMA shares its issue
    // bandwidth with ordinary VALU instructions (measured: ~6 cycles of MFMA time lost per VALU
    // instruction even with two waves per SIMD), so the steady-state loop is MFMA + LDS reads +
    // LDS-DMA + scalar bookkeeping only.
    const int w = __builtin_amdgcn_readfirstlane(wid);  // wave id, provably uniform
    unsigned offA[TA::PER_THREAD], offB[TB::PER_THREAD];  // per-lane byte offsets of the DMA pieces
#pragma unroll
    for (int q = 0; q < TA::PER_THREAD; ++q) {
      int r, k;
      TA::decode((q * 4 + w) * 64 + lane, r, k);
      offA[q] = (ALAY == GPX_LAY_KC ? (unsigned)r * (unsigned)g.lda + k : (unsigned)k * (unsigned)g.lda + r) * 8u;
    }
#pragma unroll
    for (int q = 0; q < TB::PER_THREAD; ++q) {
      int r, k;
      TB::decode((q * 4 + w) * 64 + lane, r, k);
      offB[q] = (BLAY == GPX_LAY_KC ? (unsigned)r * (unsigned)g.ldb + k : (unsigned)k * (unsigned)g.ldb + r) * 8u;
    }
    // wave-uniform byte address of the k-tile being staged, and its advance per k-tile
    // (made provably uniform once: everything derived from them below stays in SGPRs)
    const char* gA = uniform_ptr(reinterpret_cast<const char*>(ALAY == GPX_LAY_KC ? g.A + (long)i0 * g.lda + kb
                                                                                  : g.A + (long)kb * g.lda + i0));
    const char* gB = uniform_ptr(reinterpret_cast<const char*>(BLAY == GPX_LAY_KC ? g.B + (long)j0 * g.ldb + kb
                                                                                  : g.B + (long)kb * g.ldb + j0));
    const long stepA = (ALAY == GPX_LAY_KC ? (long)BK : (long)BK * g.lda) * 8;
    const long stepB = (BLAY == GPX_LAY_KC ? (long)BK : (long)BK * g.ldb) * 8;
    double* const smA = smem;
    double* const smB = smem + 2 * TA::ELEMS;
    // per-lane fragment pointers; the tile / k-step / buffer parts are immediates
    const double* pa[TA::NBASE];
    const double* pb[TB::NBASE];
#pragma unroll
    for (int i = 0; i < TA::NBASE; ++i) pa[i] = smA + TA::frag_base(i, i16, g4);
#pragma unroll
    for (int i = 0; i < TB::NBASE; ++i) pb[i] = smB + TB::frag_base(i, i16, g4);

    auto stage_a = [&](int k0, int buf) {
      if (TA::needs_guard(i0, g.M, k0, ke, g.a_tri)) {
        TA::guarded(g.A, g.lda, i0, g.M, k0, ke, g.a_tri, smA + buf * TA::ELEMS, t);
      } else {
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(gA);
#pragma unroll
        for (int q = 0; q < TA::PER_THREAD; ++q) dma16<ALAY>(gA, ra, offA[q], smA + buf * TA::ELEMS + (q * 4 + w) * 128);
      }
    };
    auto stage_b = [&](int k0, int buf) {
      if (TB::needs_guard(j0, g.N, k0, ke, g.b_tri)) {
        TB::guarded(g.B, g.ldb, j0, g.N, k0, ke, g.b_tri, smB + buf * TB::ELEMS, t);
      } else {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(gB);
#pragma unroll
        for (int q = 0; q < TB::PER_THREAD; ++q) dma16<BLAY>(gB, rb, offB[q], smB + buf * TB::ELEMS + (q * 4 + w) * 128);
      }
    };

    // Software pipeline over k-tiles, one barrier per tile placed before the LAST MFMA k-step: by then
    // this wave has read its last fragments of the current buffer (so the buffer may be overwritten)
    // and the DMA of tile t+1, issued one tile earlier, has landed.  After the barrier the DMA of tile
    // t+2 (into the buffer just released) and the first fragments of tile t+1 are issued BETWEEN the
    // MFMAs of the last k-step, so neither costs MFMA issue time and no LDS round trip is exposed at
    // the tile boundary.
    constexpr int NMFMA = TM * TN, NPIECE = TA::PER_THREAD + TB::PER_THREAD, KL = BK / 4 - 1;
    stage_a(kb, 0);
    stage_b(kb, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kb + BK < ke) {
      gA += stepA;
      gB += stepB;
      stage_a(kb + BK, 1);
      stage_b(kb + BK, 1);
    }
    double fa[2][TM], fb[2][TN];
#pragma unroll
    for (int x = 0; x < TM; ++x) fa[0][x] = pa[TA::frag_sel(wr * TM + x, 0)][TA::frag_off(wr * TM + x, 0)];
#pragma unroll
    for (int y = 0; y < TN; ++y) fb[0][y] = pb[TB::frag_sel(wc * TN + y, 0)][TB::frag_off(wc * TN + y, 0)];
    for (int k0 = kb; k0 < ke; k0 += 2 * BK) {
#pragma unroll
      for (int cur = 0; cur < 2; ++cur) {  // unrolled: the LDS buffer index is a compile-time constant
        const int kt = k0 + cur * BK;
        if (kt >= ke) break;
#pragma unroll
        for (int kk = 0; kk < KL; ++kk) {
#pragma unroll
          for (int x = 0; x < TM; ++x)
            fa[(kk + 1) & 1][x] =
                pa[TA::frag_sel(wr * TM + x, kk + 1)][cur * TA::ELEMS + TA::frag_off(wr * TM + x, kk + 1)];
#pragma unroll
          for (int y = 0; y < TN; ++y)
            fb[(kk + 1) & 1][y] =
                pb[TB::frag_sel(wc * TN + y, kk + 1)][cur * TB::ELEMS + TB::frag_off(wc * TN + y, kk + 1)];
#pragma unroll
          for (int x = 0; x < TM; ++x)
#pragma unroll
            for (int y = 0; y < TN; ++y)
              acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][x], fb[kk & 1][y], acc[x][y], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (pace != nullptr && cur == 0 && kt > kb && (((kt - kb) / BK) & (g.sync_kt - 1)) == 0) {
          if (t == 0) {
            const unsigned target = pace_live * (unsigned)((kt - kb) / (BK * g.sync_kt));
            __hip_atomic_fetch_add(pace, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int polls = 0; polls < SYNC_MAX_POLLS; ++polls) {
              if (__hip_atomic_load(pace, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
              __builtin_amdgcn_s_sleep(8);
            }
          }
        }
        __syncthreads();
        const int kn = kt + 2 * BK;
        const bool more = kn < ke;
        const bool fast = more && !TA::needs_guard(i0, g.M, kn, ke, g.a_tri) &&
                          !TB::needs_guard(j0, g.N, kn, ke, g.b_tri);
        if (more) {
          gA += stepA;
          gB += stepB;
          if (!fast) {       // boundary tile: guarded loader, ahead of the MFMAs
            stage_a(kn, cur);
            stage_b(kn, cur);
          }
        }
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(gA), rb = make_rsrc(gB);   // scalar moves only
#pragma unroll
        for (int m = 0; m < NMFMA; ++m) {
          if (fast) {   // wave-uniform: piece q of the DMA goes in front of MFMA q * NMFMA / NPIECE
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) {
              if (q * NMFMA / NPIECE == m) {
                if (q < TA::PER_THREAD) {
                  dma16<ALAY>(gA, ra, offA[q < TA::PER_THREAD ? q : 0], smA + cur * TA::ELEMS + (q * 4 + w) * 128);
                } else {
                  dma16<BLAY>(gB, rb, offB[q >= TA::PER_THREAD ? q - TA::PER_THREAD : 0],
                         smB + cur * TB::ELEMS + ((q - TA::PER_THREAD) * 4 + w) * 128);
                }
              }
            }
          }
          // fragment m of tile t+1's first k-step (of nothing useful after the last tile)
          if (m < TM)
            fa[0][m < TM ? m : 0] =
                pa[TA::frag_sel(wr * TM + m, 0)][(cur ^ 1) * TA::ELEMS + TA::frag_off(wr * TM + m, 0)];
          else if (m < TM + TN)
            fb[0][m - TM < TN ? m - TM : 0] =
                pb[TB::frag_sel(wc * TN + m - TM, 0)][(cur ^ 1) * TB::ELEMS + TB::frag_off(wc * TN + m - TM, 0)];
          acc[m / TN][m % TN] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[KL & 1][m / TN], fb[KL & 1][m % TN],
                                                                    acc[m / TN][m % TN], 0, 0, 0);
        }
#pragma unroll
        for (int f = NMFMA; f < TM + TN; ++f) {  // small tiles: fragments that did not fit between the MFMAs
          if (f < TM)
            fa[0][f < TM ? f : 0] =
                pa[TA::frag_sel(wr * TM + f, 0)][(cur ^ 1) * TA::ELEMS + TA::frag_off(wr * TM + f, 0)];
          else
            fb[0][f - TM < TN ? f - TM : 0] =
                pb[TB::frag_sel(wc * TN + f - TM, 0)][(cur ^ 1) * TB::ELEMS + TB::frag_off(wc * TN + f - TM, 0)];
        }
      }
    }
  }

  // epilogue: C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg.
  // Cin is read with clamped (always valid) addresses, 4 * TN loads in flight per step, so the
  // loads are branch-free and never serialise on their latency; stores are predicated.
  if (g.split_s > 1) {   // raw accumulators of this K slice -> workspace tile (BM x BN, row-major)
    double* wt = g.ws + (size_t)blockIdx.x * (BM * BN);
#pragma unroll
    for (int x = 0; x < TM; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int y = 0; y < TN; ++y)
          wt[((wr * TM + x) * 16 + g4 + 4 * r) * BN + (wc * TN + y) * 16 + i16] = acc[x][y][r];
    return;
  }
  const double alpha = g.alpha, beta = g.beta;
  const bool use_c = beta != 0.0;
#pragma unroll
  for (int x = 0; x < TM; ++x) {
    double cv[4][TN];
    if (use_c) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowc = min(i0 + (wr * TM + x) * 16 + g4 + 4 * r, g.M - 1);
#pragma unroll
        for (int y = 0; y < TN; ++y) {
          const int colc = min(j0 + (wc * TN + y) * 16 + i16, g.N - 1);
          cv[r][y] = g.Cin[(long)rowc * g.ldcin + colc];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + (wr * TM + x) * 16 + g4 + 4 * r;
#pragma unroll
      for (int y = 0; y < TN; ++y) {
        const int col = j0 + (wc * TN + y) * 16 + i16;
        double v = alpha * acc[x][y][r];
        if (use_c) v += beta * cv[r][y];
        if (row == col) v += g.diag_add;
        bool ok = row < g.M && col < g.N;
        if (g.c_upper) {
          if (row == col && ok) g.diag_out[row] = v;
          ok = ok && col > row;
        } else if (g.c_lower) {
          ok = ok && col <= row;
        }
        if (ok) g.D[(long)row * g.ldd + col] = v;
      }
    }
  }
}

// Is slot `within` (0..63) of super-tile st a live (non-empty, on-or-below-diagonal) tile?  Host mirror
// of block_coords' super-tile mapping.
bool tile_live(unsigned st, unsigned within, int tiles_m, int tiles_n, int lower, int order) {
  unsigned sr, sc;
  if (lower) {
    sr = (unsigned)((sqrt(8.0 * (double)st + 1.0) - 1.0) * 0.5);
    while ((sr + 1) * (sr + 2) / 2 <= st) ++sr;
    while (sr * (sr + 1) / 2 > st) --sr;
    sc = st - sr * (sr + 1) / 2;
  } else {
    super_coords(st, tiles_m, tiles_n, order, sr, sc);
  }
  const int bi = (int)(sr * 8 + (within >> 3)), bj = (int)(sc * 8 + (within & 7));
  return bi < tiles_m && bj < tiles_n && (!lower || bj <= bi);
}

// Sum of the split_s partial tiles of each tail tile (fixed order p = 0, 1, ...) and the GEMM epilogue
// D = alpha * sum + beta * Cin (+ diag_add), one workgroup per tile (packed tile ids from tile0 on).
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmDesc g, int tiles_m, int tiles_n, unsigned tile0) {
  // 16 workgroups per tile, 4 elements per thread, the split_s loads of an element independent of each
  // other (one workgroup per tile walked 1024 dependent load rounds: 240 us for 16 tiles)
  constexpr int CHUNKS = 16, PER = BM * BN / CHUNKS / 256;
  const unsigned tile = blockIdx.x / CHUNKS, chunk = blockIdx.x % CHUNKS;
  int bi, bj;
  if (!block_coords(tiles_m, tiles_n, g.c_lower, 0, tile0 + tile, 0, bi, bj)) return;
  const double* w0 = g.ws + (size_t)tile * g.split_s * (BM * BN);
  double v[PER];
  int e[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    e[i] = (int)chunk * (BM * BN / CHUNKS) + i * 256 + threadIdx.x;
    v[i] = 0.0;
  }
#pragma unroll 4
  for (int p = 0; p < g.split_s; ++p) {
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] += w0[(size_t)p * (BM * BN) + e[i]];
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int row = bi * BM + e[i] / BN, col = bj * BN + e[i] % BN;
    if (row >= g.M || col >= g.N || (g.c_lower && col > row)) continue;
    double r = v[i] * g.alpha;
    if (g.beta != 0.0) r += g.beta * g.Cin[(long)row * g.ldcin + col];
    if (row == col) r += g.diag_add;
    g.D[(long)row * g.ldd + col] = r;
  }
}

thread_local double* t_ws = nullptr;
thread_local size_t t_ws_elems = 0;

long env_long(const char* name, long dflt) {
  const char* e = getenv(name);
  return e ? atol(e) : dflt;
}

constexpr unsigned TAIL_DYN_LDS = 32768;   // 64 KiB static + 32 KiB: two workgroups no longer fit in a CU's 160 KiB

bool tail_split_enabled() {
  static const bool on = [] {
    const char* e = getenv("GPX_GEMM_TAIL");   // development switch: 0 disables the separate tail launch
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

template <int TM, int TN>
int launch_cfg(hipStream_t s, const GemmDesc& g) {
  constexpr int BM = 32 * TM, BN = 32 * TN;
  const int tm = (g.M + BM - 1) / BM, tn = (g.N + BN - 1) / BN;
  unsigned nsuper;
  if (g.c_lower) {
    const unsigned S = (unsigned)((max(tm, tn) + 7) / 8);
    nsuper = S * (S + 1) / 2;
  } else {
    nsuper = (unsigned)((tm + 7) / 8) * (unsigned)((tn + 7) / 8);
  }
  dim3 grid(((nsuper + 7) / 8) * 8 * 64), block(NTHREADS);
  // Launches that cannot fill the chip anyway (at most two workgroups per CU) skip the XCD
  // super-tile mapping: one block per tile, so the dispatcher spreads them over all XCDs and CUs
  // instead of packing the few live blocks of a sparse grid onto a handful of CUs.
  const long ntiles = g.c_lower ? (long)max(tm, tn) * (max(tm, tn) + 1) / 2 : (long)tm * tn;
  // Mid-size LOWER launches too: the triangular grid of 8 x 8 super-tiles deals the XCDs unequal
  // numbers of live tiles (32 x 32 tiles: 100 on two XCDs, 36 on two others -> two rounds for one
  // round of work); packed row-major ids go round-robin over the XCDs -- every XCD then holds every
  // 8th tile of a few rows, the same 8 + 8 panels per 64 tiles as a super-tile once rows are 64 wide.
  static const long packed_max = env_long("GPX_GEMM_PACKED_MAX", 9000);   // development override
  const bool packed = ntiles <= 512 || (g.c_lower && TM * TN >= 16 && ntiles <= packed_max);
  if (packed) {
    nsuper = 0;
    grid = dim3((unsigned)ntiles);
  }
  const bool batched = g.batch > 1;   // grid.y: no separate tail, no split-K, no pacing (one launch, one part)
  if (batched) grid.y = (unsigned)g.batch;
  // heavy tiles first (see super_coords): which side is heavy follows from how gemm_f64_kernel clips k
  int order = ORD_ROWS_SKEWED;
  static const bool lpt = getenv("GPX_GEMM_LPT") == nullptr || atoi(getenv("GPX_GEMM_LPT")) != 0;   // development switch
  if (!g.c_lower && lpt) {
    if (g.b_tri) order = g.b_lay == GPX_LAY_RC ? ORD_COLS_ASC : ORD_COLS_DESC;      // k >= j0  |  k < j0 + BN
    else if (g.a_tri && g.a_lay == GPX_LAY_KC) order = ORD_ROWS_DESC;               // k < i0 + BM
    else if (g.a_tri) order = ORD_ROWS_ASC_TRI;   // k >= i0: heavy rows come first anyway; marks the operand triangular
  }
  // lower launches are enumerated row by row; with a triangular operand the work per tile changes monotonically
  // along that order (lauum: k >= i0), which is all the one-round pairing needs to know
  if (g.c_lower && lpt && (g.a_tri || g.b_tri)) order = ORD_ROWS_ASC_TRI;
  static const bool log_shapes = getenv("GPX_GEMM_LOG") != nullptr;   // development: one line per kernel launch
  auto log_launch = [&](const char* part, unsigned blocks) {
    if (log_shapes)
      fprintf(stderr, "GEMMLOG %d %d %d lower=%d atri=%d btri=%d inplace=%d cfg=%d part=%s blocks=%u cb=%d\n", g.M, g.N, g.K,
              g.c_lower, g.a_tri, g.b_tri, g.inplace, TM * 10 + TN, part, blocks, g.cb_off ? g.cb_nb : 0);
  };
  // pacing counters (one per super-tile, at the head of the workspace) for the main part of a long-K
  // launch in super-tile mode with plain operands
  // OFF by default: pacing lifts the L2 hit rate of the dominant launch from 36 % to 86 % (fabric reads
  // 4.8x lower) but not its speed -- the misses are served by the Infinity Cache / HBM at 1.5 TB/s, far from
  // binding, while waiting for the slowest partner costs 0.3-1 % (M = N = K = 16384: 73.2 unpaced, 73.0 /
  // 72.6 / 72.2 TFLOP/s paced every 128 / 64 / 32 k-tiles; potrf(65536) 1365 -> 1378 ms).  Kept as a switch
  // because it is the measured answer to "why is the fabric traffic 20x the algorithmic bytes".
  static const long pace_on = env_long("GPX_GEMM_PACE", 0);   // development switch
  static const long pace_kt = env_long("GPX_GEMM_PACE_KT", SYNC_KT);   // development override (a power of two)
  GemmDesc gp = g;
  if (pace_on && !batched && nsuper != 0 && TM * TN >= 16 && t_ws != nullptr && !g.a_tri && !g.b_tri && !g.inplace && !g.cb_off &&
      g.K >= 2 * pace_kt * BK && (size_t)nsuper * sizeof(unsigned) <= t_ws_elems * sizeof(double)) {
    gp.sync_cnt = reinterpret_cast<unsigned*>(t_ws);
    gp.sync_kt = (int)pace_kt;
    if (hipMemsetAsync(gp.sync_cnt, 0, (size_t)nsuper * sizeof(unsigned), s) != hipSuccess) gp.sync_cnt = nullptr;
  }
  auto launch = [&](dim3 gr, unsigned dyn, unsigned bid0) {
    log_launch(dyn ? "tail" : "main", gr.x);
    const GemmDesc& gk = dyn ? g : gp;   // (the tail runs one workgroup per CU, unpaced)
    if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_KC)
      hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_KC, TM, TN>), gr, block, dyn, s, gk, tm, tn, nsuper, bid0, order);
    else if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_RC)
      hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_RC, TM, TN>), gr, block, dyn, s, gk, tm, tn, nsuper, bid0, order);
    else if (g.a_lay == GPX_LAY_RC && g.b_lay == GPX_LAY_RC)
      hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_RC, TM, TN>), gr, block, dyn, s, gk, tm, tn, nsuper, bid0, order);
    else
      hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_KC, TM, TN>), gr, block, dyn, s, gk, tm, tn, nsuper, bid0, order);
  };
  // Tail of a multi-round launch.  Workgroups are dispatched in id order, 512 live ones per round of
  // the chip.  When the last round(s) hold few live tiles, the slots they land in are the
  // two halves of the CUs that finish first, so the tail runs two workgroups per CU on a few CUs --
  // at half speed each, the fp64 MFMA pipe being shared -- while the other CUs idle (measured: a
  // 32-tile tail cost 0.43 ms against 0.16 ms for the same tiles one per CU).  Such a tail is
  // launched separately with enough dynamic LDS that only ONE workgroup fits per CU.
  unsigned main_ids = grid.x;
  const long rem = ntiles % 512;   // live tiles beyond the last full round
  if (!batched && nsuper != 0 && TM * TN >= 16 && rem > 0 && rem <= 256 && tail_split_enabled()) {
    // whole trailing 512-id chunks (8 super-tiles) only: a finer cut was measured and lost -- where the
    // main part does not end on a chunk boundary its workgroups do not finish together, and the tail
    // overlaps with them better inside one launch
    long live = 0;
    unsigned chunk = grid.x / 512;
    while (chunk > 1 && live < rem) {
      --chunk;
      for (unsigned x = 0; x < 8; ++x)
        for (unsigned within = 0; within < 64; ++within)
          if (chunk * 8 + x < nsuper && tile_live(chunk * 8 + x, within, tm, tn, g.c_lower, order)) ++live;
    }
    if (live == rem) main_ids = chunk * 512;
  }
  if (!batched && packed && ntiles > 512 && rem > 0 && rem <= 256 && tail_split_enabled()) {
    // packed ids: the tail is simply the last `rem` tiles.  Few of them and a long K: split K over
    // several workgroups per tile (one workgroup per CU, all CUs busy, each at full MFMA rate), partial
    // tiles into the workspace, summed in a fixed order.  Otherwise one workgroup per CU as below.
    static const long splitk_on = env_long("GPX_GEMM_SPLITK", 1);   // development switch
    const unsigned main_tiles = (unsigned)(ntiles - rem);
    launch(dim3(main_tiles), 0, 0);
    int split = 0;
    if (splitk_on && t_ws != nullptr && rem <= 128 && g.K >= 1024 && !g.a_tri && !g.b_tri && !g.inplace && !g.c_upper && !g.cb_off) {
      split = (int)(256 / rem);
      while (split > 1 && g.K / split < 128) --split;
      if ((size_t)rem * split * (BM * BN) > t_ws_elems) split = (int)(t_ws_elems / ((size_t)rem * BM * BN));
    }
    if (split > 1) {
      GemmDesc gs = g;
      gs.split_s = split;
      gs.split_tile0 = main_tiles;
      gs.ws = t_ws;
      auto launch_split = [&](dim3 gr) {
        log_launch("split", gr.x);
        if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_KC)
          hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_KC, TM, TN>), gr, block, TAIL_DYN_LDS, s, gs, tm, tn, nsuper, 0u, order);
        else if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_RC)
          hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_RC, TM, TN>), gr, block, TAIL_DYN_LDS, s, gs, tm, tn, nsuper, 0u, order);
        else if (g.a_lay == GPX_LAY_RC && g.b_lay == GPX_LAY_RC)
          hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_RC, TM, TN>), gr, block, TAIL_DYN_LDS, s, gs, tm, tn, nsuper, 0u, order);
        else
          hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_KC, TM, TN>), gr, block, TAIL_DYN_LDS, s, gs, tm, tn, nsuper, 0u, order);
      };
      launch_split(dim3((unsigned)(rem * split)));
      hipLaunchKernelGGL((splitk_reduce_kernel<BM, BN>), dim3((unsigned)rem * 16), dim3(256), 0, s, gs, tm, tn, main_tiles);
    } else {
      launch(dim3((unsigned)rem), TAIL_DYN_LDS, main_tiles);
    }
  } else if (main_ids == grid.x) {
    launch(grid, 0, 0);
  } else {
    launch(dim3(main_ids), 0, 0);
    launch(dim3(grid.x - main_ids), TAIL_DYN_LDS, main_ids);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

}  // namespace

void gpx_gemm_set_workspace(double* ws, size_t elems) {
  t_ws = ws;
  t_ws_elems = elems;
}

int gpx_gemm_launch(hipStream_t s, const GemmDesc& g) {
  if (g.M <= 0 || g.N <= 0) return 0;
  GPX_REQUIRE(g.K >= 0, "gemm: negative K");
  GPX_REQUIRE(g.batch >= 1 && g.batch <= 65535, "gemm: batch out of range");
  GPX_REQUIRE(g.batch == 1 || (!g.inplace && !g.c_upper && g.split_s <= 1), "gemm: batched launches are plain products");
  GPX_REQUIRE(g.A && g.B && g.D, "gemm: null operand");
  GPX_REQUIRE(g.beta == 0.0 || g.Cin, "gemm: beta != 0 needs Cin");
  GPX_REQUIRE(!g.c_upper || (g.c_lower && g.diag_out && g.M == g.N && g.beta == 0.0),
              "gemm: c_upper needs c_lower, a diagonal vector, a square product and beta = 0");
  GPX_REQUIRE(gpx_aligned16(g.A) && gpx_aligned16(g.B), "gemm: operands must be 16-byte aligned");
  GPX_REQUIRE((g.lda & 1) == 0 && (g.ldb & 1) == 0, "gemm: leading dimensions must be even");
  GPX_REQUIRE(g.lda >= (g.a_lay == GPX_LAY_KC ? g.K : g.M), "gemm: lda too small");
  GPX_REQUIRE(g.ldb >= (g.b_lay == GPX_LAY_KC ? g.K : g.N), "gemm: ldb too small");
  GPX_REQUIRE(g.cb_off != nullptr || g.ldd >= g.N, "gemm: ldd too small");
  GPX_REQUIRE(g.cb_off == nullptr || (g.cb_row0 != nullptr && g.cb_nb >= 128 && g.cb_nb % 128 == 0 && g.cb_rowbase % 128 == 0 &&
                                       g.ldd == g.cb_nb && g.ldcin == g.cb_nb && g.Cin == g.D && !g.c_lower && !g.inplace &&
                                       g.batch == 1),
              "gemm: column-block output needs tables, a block width / row base that are multiples of 128, ldd = the block width and Cin = D");
  GPX_REQUIRE(g.lda < (1L << 21) && g.ldb < (1L << 21), "gemm: leading dimension >= 2^21 (32-bit lane offsets)");
  // Very long K: the 64 workgroups of a super-tile run without any mutual synchronisation and drift
  // apart in k until the panels they share no longer meet in the XCD's 4 MiB L2 (measured, M = N = K =
  // 32768: 71.1 TFLOP/s in one launch, 72.3 as four launches of K = 8192 -- the launch boundary
  // re-aligns them; the extra pass over C costs 2 ms of HBM time per boundary, under the MFMAs).
  // Plain operands only: a triangular operand's k range is tied to the tile position.
  static const long kchunk = [] {
    const char* e = getenv("GPX_GEMM_KCHUNK");   // development override; 0 disables
    return e ? atol(e) : 8192L;
  }();
  if (kchunk > 0 && g.batch <= 1 && g.K >= 2 * kchunk && !g.a_tri && !g.b_tri && !g.inplace && !g.c_upper && g.split_s <= 1 &&
      (long)g.M * g.N >= 4096L * 4096L) {
    GemmDesc c = g;
    for (int k0 = 0; k0 < g.K; k0 += c.K) {
      c.K = g.K - k0 < 2 * kchunk ? g.K - k0 : (int)kchunk;   // the last chunk takes the remainder (< 2 chunks: no recursion)
      c.A = g.a_lay == GPX_LAY_KC ? g.A + k0 : g.A + (long)k0 * g.lda;
      c.B = g.b_lay == GPX_LAY_KC ? g.B + k0 : g.B + (long)k0 * g.ldb;
      if (k0 > 0) { c.Cin = g.D; c.ldcin = g.ldd; c.beta = 1.0; c.diag_add = 0.0; }
      const int rc = gpx_gemm_launch(s, c);
      if (rc != 0) return rc;
    }
    return 0;
  }
  // 128 x 128 tiles unless that leaves most of the 256 CUs (2 workgroups each) idle.  In-place
  // launches (D aliases an operand) rely on one workgroup owning a whole operand panel: 128 only.
  const long big_tiles = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) / (g.c_lower ? 2 : 1) * (g.batch > 1 ? g.batch : 1);
  // In-place right-multiplies (D aliases A, N <= 128) with few rows: 32-row tiles, still 128 wide, so
  // that a 128 x 128 x 128 leaf solve is spread over 4 CUs instead of being one CU's 13.7 us of MFMA
  // time.  (When D aliases B the workgroup must own all K = M <= 128 rows: 128-row tiles only.)
  if (g.inplace == GPX_INPLACE_A && g.M <= 16384) return launch_cfg<1, 4>(s, g);
  // The mirror image for in-place left-multiplies (D aliases B, M = K <= 128: the leaves of the left
  // solves): 128 x 32 tiles, the workgroup still owns all 128 rows of its column panel.
  if (g.inplace == GPX_INPLACE_B && g.N <= 16384) return launch_cfg<4, 1>(s, g);
  if (g.inplace || big_tiles >= 384) return launch_cfg<4, 4>(s, g);
  return launch_cfg<2, 2>(s, g);
}
