// blokus.hip -- batched Blokus (20x20, 4 players) for gfx950 (MI355X).  Hand-written HIP, wave64.
//
// Restates, for B independent games at once:
//   colosseumrl/envs/blokus/board.py:170-193        get_all_valid_moves (order: piece -> anchor -> orientation -> shift)
//   colosseumrl/envs/blokus/board.py:114-154        anchor ("corner") cells
//   colosseumrl/envs/blokus/computation.py:122-246  cell legality, orientations, shift ids
//   colosseumrl/envs/blokus/BlokusEnvironment.py:357-451  next_state (terminal test on the PRE-move board)
//   colosseumrl/envs/blokus/ai.py:44-54             inventory / score / last-piece bonus
//
// Mapping: ONE WAVE PER GAME.  The board lives in LDS as 4 colours x 20 row words (bit x = column x).
// For a player q two row bitboards are derived once per step,
//     allowed_q = empty & ~orth(own_q)                 (a piece may only cover such cells)
//     corner_q  = allowed_q & diag(own_q)              (anchors; round 0: the player's board corner)
// and the reference's triple loop over (anchor, orientation, shift) collapses to whole-shape fitting:
// for an oriented shape S with cells s_j, the origins where it fits are F = AND_j (allowed >> s_j)
// (K-1 shift-ANDs per board row), and the number of reference actions it contributes is
// sum_j popcount(F & (corner >> s_j)) -- one action per (cell of S lying on an anchor).
// Lanes own (piece, orientation) pairs; 8 consecutive lanes make up one piece.  The random agent's
// "r-th legal action in reference order" is found by a three-level search (piece totals -> anchors of
// that piece, one lane per anchor, wave prefix scan -> the r-th (orientation, shift) bit of that anchor).
// The 168 oriented shapes (21 pieces x 8 orientations, 5 packed cell bytes each) sit in LDS.
#include "crl_common.hpp"
#include <type_traits>

namespace {

// waves per SIMD the register allocator must leave room for.  Left alone it takes 256 VGPRs (1 wave/SIMD, the
// 16,384 one-wave games of BASELINE config 4 then run in 16 rounds).  Measured on MI355X (-DBLK_WAVES_PER_SIMD=n variants, tools/lib_variant.sh):
// 4/5/6/8 waves per SIMD -> 244/237/256/265 M env-steps/s; 8 (64 VGPRs + 96 B of spills) wins.
#ifndef BLK_WAVES_PER_SIMD
#define BLK_WAVES_PER_SIMD 8
#endif

constexpr int BN = 20;
constexpr int NPIECE = 21;
constexpr int NSHAPE = NPIECE * 8;
constexpr int NDISTINCT = 91;                             // distinct oriented shapes over all 21 pieces (build_distinct checks)
constexpr uint32_t ROWMASK = (1u << BN) - 1u;
constexpr int ACTION_IDS = NPIECE * 400 * 8 * 5;          // 336,000 dense ids
constexpr int MASK_WORDS = ACTION_IDS / 32;               // 10,500

// (dx, dy) cell offsets, (0,0) first; order = inventory order = action order (board.py:24-44)
const int8_t kPieces[NPIECE][5][2] = {
    {{0, 0}},
    {{0, 0}, {1, 0}},
    {{0, 0}, {1, 0}, {1, 1}},
    {{0, 0}, {1, 0}, {2, 0}},
    {{0, 0}, {1, 0}, {0, 1}, {1, 1}},
    {{0, 0}, {1, -1}, {1, 0}, {2, 0}},
    {{0, 0}, {1, 0}, {2, 0}, {3, 0}},
    {{0, 0}, {1, 0}, {2, 0}, {2, -1}},
    {{0, 0}, {1, 0}, {1, -1}, {2, -1}},
    {{0, 0}, {0, -1}, {1, 0}, {2, 0}, {3, 0}},
    {{0, 0}, {0, -1}, {0, 1}, {1, 0}, {2, 0}},
    {{0, 0}, {0, -1}, {0, -2}, {1, -2}, {2, -2}},
    {{0, 0}, {1, 0}, {1, -1}, {2, -1}, {3, -1}},
    {{0, 0}, {0, 1}, {1, 0}, {2, 0}, {2, -1}},
    {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}},
    {{0, 0}, {1, 0}, {2, 0}, {1, -1}, {2, -1}},
    {{0, 0}, {0, 1}, {1, 0}, {1, -1}, {2, -1}},
    {{0, 0}, {1, 0}, {0, 1}, {0, 2}, {1, 2}},
    {{0, 0}, {1, 0}, {1, -1}, {1, 1}, {2, -1}},
    {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}},
    {{0, 0}, {1, 0}, {1, -1}, {2, 0}, {3, 0}},
};
constexpr int8_t kPieceCells[NPIECE] = {1, 2, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5};

// device-resident tables, staged into LDS by every workgroup
struct BlkTables {
    uint8_t cells[NSHAPE][8];   // [piece*8+orient][j] = (dx+4) | (dy+4) << 4 ; entries j >= n repeat cell 0
    uint8_t ncell[24];          // cells per piece (= its score value, ai.py:12-22)
    uint8_t nuniq[24];          // distinct oriented shapes of the piece (1..8; 91 over all 21 pieces)
    uint8_t uniq[NPIECE][8];    // k-th distinct orientation: orient | multiplicity << 4.  Orientations that give the
                                // same cell set (symmetric pieces) yield the same NUMBER of legal actions, so the
                                // count pass fits each distinct shape once and multiplies.
    // The 91 distinct oriented shapes as the count / existence passes want them (round 3): per cell the byte offset of its
    // row 0 in the wave's pre-shifted table (WaveLds::u.sh[dy+4][dx+4]) relative to the start of WaveLds -- the slots behind
    // the shape's cells repeat cell 0, see blk_shape_count --, multiplicity | (cells - 1) << 4, and the piece: one 16-byte
    // read per work item where round 2 went item -> uniq -> cells / ncell and then computed five addresses (~45 VALU per batch).
    // `rows`: the least and the greatest row offset dy + 4 of the shape's cells (bits 3:0, 7:4).  The table is ordered by
    // the shapes' extent in rows (greatest - least), flat shapes first -- the order of the work list, so that the shapes of
    // a batch need about the same number of origin rows (blk_count_batches).
    struct alignas(16) Distinct { uint16_t off[5]; uint8_t mult_n, piece; uint32_t rows; } distinct[92];
    uint8_t first[24];          // index of a piece's first distinct shape in `distinct`
};

// orientation o of offset (dx,dy): ORIENTATIONS order of board.py:47, maps of computation.py:54-86
void orient_offset(int o, int dx, int dy, int &ox, int &oy)
{
    switch (o) {
        case 0: ox = dy; oy = -dx; break;    // north      (270 deg)
        case 1: ox = dx; oy = -dy; break;    // northeast  (flip y)
        case 2: ox = dx; oy = dy; break;     // east       (identity)
        case 3: ox = dy; oy = dx; break;     // southeast  (90 deg, flip x)
        case 4: ox = -dy; oy = dx; break;    // south      (90 deg)
        case 5: ox = -dx; oy = dy; break;    // southwest  (180 deg, flip y)
        case 6: ox = -dx; oy = -dy; break;   // west       (180 deg)
        default: ox = -dy; oy = -dx; break;  // northwest  (270 deg, flip x)
    }
}

void build_tables(BlkTables &t)
{
    memset(&t, 0, sizeof(t));
    for (int p = 0; p < NPIECE; ++p) {
        t.ncell[p] = (uint8_t)kPieceCells[p];
        for (int o = 0; o < 8; ++o)
            for (int j = 0; j < 8; ++j) {
                const int jj = j < kPieceCells[p] ? j : 0;
                int ox, oy;
                orient_offset(o, kPieces[p][jj][0], kPieces[p][jj][1], ox, oy);
                t.cells[p * 8 + o][j] = (uint8_t)((ox + 4) | ((oy + 4) << 4));
            }
        // group the 8 orientations by their cell set (translation-normalised, order-independent)
        uint32_t key[8][5];
        for (int o = 0; o < 8; ++o) {
            int xs[5], ys[5], mx = 99, my = 99;
            const int n = kPieceCells[p];
            for (int j = 0; j < n; ++j) {
                orient_offset(o, kPieces[p][j][0], kPieces[p][j][1], xs[j], ys[j]);
                mx = xs[j] < mx ? xs[j] : mx;
                my = ys[j] < my ? ys[j] : my;
            }
            for (int j = 0; j < 5; ++j) key[o][j] = j < n ? (uint32_t)((ys[j] - my) * 16 + (xs[j] - mx)) : 999u;
            for (int a = 0; a < 5; ++a)
                for (int b2 = a + 1; b2 < 5; ++b2)
                    if (key[o][b2] < key[o][a]) { const uint32_t tmp = key[o][a]; key[o][a] = key[o][b2]; key[o][b2] = tmp; }
        }
        int nu = 0;
        for (int o = 0; o < 8; ++o) {
            int same = -1;
            for (int k = 0; k < nu && same < 0; ++k)
                if (memcmp(key[t.uniq[p][k] & 7], key[o], sizeof(key[o])) == 0) same = k;
            if (same >= 0) t.uniq[p][same] = (uint8_t)(t.uniq[p][same] + 16);
            else t.uniq[p][nu++] = (uint8_t)(o | (1 << 4));
        }
        t.nuniq[p] = (uint8_t)nu;
    }
}

// The pre-shifted table of the count pass: BLK_SH_ROWS rows (board rows -4 .. 26; only 0..19 are not zero) of 9 entries, one
// per column offset a shape cell can have -- ROW-major since the second half of round 3.  Which (shift, row) pairs of a
// shape x row loop share an LDS bank is what the layout decides, and the count pass is bound by the LDS array: modelled
// on random inventories (the 64 lanes' five reads per origin row, `ds_read_b64` banking), shift-major with 28 rows per shift
// -- the best of the shift-major strides, and what rounds 2 / 3 measured as fastest -- takes 1.39 LDS cycles per conflict-free
// cycle, row-major with 9 entries per row 1.21 (10: 1.33, 11: 1.49, 12: 1.27).
// Rows 28..30 stay zero for the whole launch: a shape that shares its origin rows out over two or four lanes runs every lane
// for the same number of rows (scalar loop control), so the last lane overshoots the row range by up to three rows.
constexpr int BLK_SH_ROWS = 31;
constexpr int BLK_SH_ROW_BYTES = 9 * (int)sizeof(uint2);

// per-wave working set in LDS
struct WaveLds {
    uint32_t occ[4][BN];     // board rows per colour, bit x = column x
    uint2 ac[4][32];         // per player, index y+4: {allowed << 8, corner << 8}; rows outside the board are 0
    struct {                 // (a union with the select pass's fit table and anchor list until the middle of round 3: both are gone)
        uint2 sh[BLK_SH_ROWS][9];   // the player being counted: sh[r][s] = ac[q][r] >> s (both words), s = dx + 4 of a shape cell;
                                    // rows 0..27 are rewritten by every count pass, rows 28..30 stay zero
    } u;
    uint32_t pcnt[32];       // legal-action count per piece (0 for pieces not held)
    uint8_t items[NSHAPE];   // work list of a count / existence pass: indices into BlkTables::distinct
};

// the `distinct` / `first` part of the tables (needs the layout of WaveLds)
void build_distinct(BlkTables &t)
{
    BlkTables::Distinct all[NSHAPE];
    int ext[NSHAPE], n = 0;
    for (int p = 0; p < NPIECE; ++p) {
        t.first[p] = (uint8_t)n;                    // (position in piece-major order: host-side bookkeeping only)
        for (int k = 0; k < t.nuniq[p]; ++k, ++n) {
            const int o = t.uniq[p][k] & 7;
            BlkTables::Distinct &e = all[n];
            memset(&e, 0, sizeof(e));
            int rmin = 8, rmax = 0;
            for (int j = 0; j < 5; ++j) {
                const int c = t.cells[p * 8 + o][j], sx = c & 15, ro = c >> 4;
                // (cells[][j] repeats cell 0 for j >= the piece's cell count: so do the table offsets)
                const size_t off = offsetof(WaveLds, u) + (size_t)(ro * 9 + sx) * sizeof(uint2);
                e.off[j] = (uint16_t)off;
                rmin = ro < rmin ? ro : rmin;
                rmax = ro > rmax ? ro : rmax;
            }
            e.mult_n = (uint8_t)((t.uniq[p][k] >> 4) | ((t.ncell[p] - 1) << 4));
            e.piece = (uint8_t)p;
            e.rows = (uint32_t)(rmin | (rmax << 4));
            ext[n] = rmax - rmin;
        }
    }
    int d = 0;
    for (int want = 0; want <= 8; ++want)           // stable: piece-major inside an extent
        for (int i = 0; i < n; ++i)
            if (ext[i] == want) t.distinct[d++] = all[i];
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int nth_set_bit32(uint32_t m, int r)
{
    int pos = 0, c;
    c = __popc(m & 0xffffu); if (r >= c) { r -= c; m >>= 16; pos += 16; }
    c = __popc(m & 0xffu);   if (r >= c) { r -= c; m >>= 8;  pos += 8; }
    c = __popc(m & 0xfu);    if (r >= c) { r -= c; m >>= 4;  pos += 4; }
    c = __popc(m & 0x3u);    if (r >= c) { r -= c; m >>= 2;  pos += 2; }
    c = (int)(m & 1u);       if (r >= c) { pos += 1; }
    return pos;
}

__device__ __forceinline__ int nth_set_bit64(unsigned long long m, int r)
{
    const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
    const int c = __popc(lo);
    return r < c ? nth_set_bit32(lo, r) : 32 + nth_set_bit32(hi, r - c);
}

// inclusive prefix sum across the 64 lanes: Hillis-Steele inside each row of 16 lanes with DPP row shifts (the shifted
// operand is folded into the add), then the row totals with the two row-broadcast steps.  8 VALU, no LDS crossbar.
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, const int lane)
{
    (void)lane;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// ---- derive allowed / corner rows from the board: of all four players (lanes 0..79: player q, row y), or of player
// `only` alone (lanes 0..19)
__device__ __forceinline__ uint2 blk_prep_row(const uint32_t (&occ)[4][BN], const int q, const int y, const int round)
{
    const uint32_t any = occ[0][y] | occ[1][y] | occ[2][y] | occ[3][y];
    const uint32_t own = occ[q][y];
    const uint32_t up = y > 0 ? occ[q][y - 1] : 0u, dn = y < BN - 1 ? occ[q][y + 1] : 0u;
    const uint32_t orth = up | dn | (own << 1) | (own >> 1);           // computation.py:89-119
    const uint32_t allowed = ~any & ~orth & ROWMASK;                  // computation.py:122-142
    uint32_t corner;
    if (round == 0) {                                                // board.py:177-179, corners of board.py:50
        const int cx = (q & 1) ? BN - 1 : 0, cy = (q & 2) ? BN - 1 : 0;
        corner = (y == cy) ? (allowed & (1u << cx)) : 0u;
    } else {                                                         // board.py:114-154
        const uint32_t ud = up | dn;
        corner = allowed & ((ud << 1) | (ud >> 1)) & ROWMASK;
    }
    return make_uint2(allowed << 8, corner << 8);
}

__device__ __forceinline__ void blk_prep(WaveLds &L, const int lane, const int round, const int only = -1)
{
    const int first = only < 0 ? 0 : only * BN, last = only < 0 ? 4 * BN : (only + 1) * BN;
    for (int i = first + lane; i < last; i += 64) {
        const int q = i / BN, y = i - q * BN;
        L.ac[q][y + 4] = blk_prep_row(L.occ, q, y, round);
    }
    wave_sync();
}

// one oriented shape in two registers: 5 cell bytes (dx+4 | dy+4 << 4), cells j >= n repeat cell 0.
// Fields are extracted where used (v_bfe): the kernels are latency-bound, registers are the scarcer resource.
struct ShapeRegs {
    uint32_t c03, c4;
    int n;
    __device__ __forceinline__ uint32_t cell(const int j) const { return j < 4 ? (c03 >> (8 * j)) & 0xffu : c4 & 0xffu; }
    __device__ __forceinline__ int sh(const int j) const { return (int)(cell(j) & 15u); }       // dx + 4: shift amount
    __device__ __forceinline__ int ro(const int j) const { return (int)(cell(j) >> 4); }        // dy + 4: row offset
    // shift for the anchor test: real cells use sh, padding cells 31 (every row shifts to 0)
    __device__ __forceinline__ int shc(const int j) const { return j < n ? sh(j) : 31; }
};

__device__ __forceinline__ ShapeRegs blk_load_shape(const BlkTables &T, const int piece, const int orient)
{
    ShapeRegs s;
    const uint2 raw = *reinterpret_cast<const uint2 *>(&T.cells[piece * 8 + orient][0]);
    s.c03 = raw.x;
    s.c4 = raw.y;
    s.n = T.ncell[piece];
    return s;
}

// Rows of player q's allowed / corner masks pre-shifted by every column offset a shape cell can have (dx + 4 = 0..8).
// A count pass tests ~90 oriented shapes x up to 20 origin rows x 5 cells; with the shifts done here, once per row and
// offset, a cell test is one 8-byte LDS read and two ANDs instead of a read, two variable shifts and two ANDs.
__device__ __forceinline__ void blk_build_shifted(WaveLds &L, const int q, const int lane)
{
    // one trip: lanes 0..27 take a row each and its shifts 0..4, lanes 32..59 the same rows' shifts 5..8 (one read, five /
    // four 8-byte writes per lane; as 252 (shift, row) items over 64 lanes it was four trips with a division each)
    const int r = lane & 31, s0 = lane < 32 ? 0 : 5;
    if (r < 28) {
        const uint2 v = L.ac[q][r];
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (s0 + k < 9) L.u.sh[r][s0 + k] = make_uint2(v.x >> (s0 + k), v.y >> (s0 + k));
    }
    wave_sync();
}

// actions contributed by one oriented shape when its origin lies in rows [y0, y1]; any_only: stop at the first hit.
// blk_build_shifted(q) must have run.
// (Tried in round 3: cell 0 of every shape is its origin, i.e. the same table row in all 64 lanes, so with wave-uniform
//  rows it can come out of a register by v_readlane instead of out of the LDS -- four reads per row instead of five.
//  7 % SLOWER: the pass is bound by instruction issue, not by the LDS, and a v_readlane with a scalar index stalls.)
struct DistinctRegs { uint32_t o01, o23, o4mp, rows; };   // BlkTables::Distinct as loaded: off[0..4], mult | (cells - 1) << 4, piece, rows

__device__ __forceinline__ DistinctRegs blk_load_distinct(const BlkTables &T, const int d)
{
    const uint4 raw = *reinterpret_cast<const uint4 *>(&T.distinct[d]);
    return DistinctRegs{raw.x, raw.y, raw.z, raw.w};
}

// Origin rows ya .. ya + n - 1: `ya` may differ from lane to lane (lanes that share a shape take a part of its rows each),
// `n` is wave-uniform -- the loop is controlled on the scalar unit.  Rows behind the range proper (at most three, and at
// most up to board row 22, i.e. table row 30) count nothing: behind the last board row the origin cell itself lies on a
// zero row of the table, and behind a range that ends four rows past the last anchor row no cell of a shape can reach an
// anchor.
// (Round 2 / early round 3 gave such lanes their own first and last row: a loop with per-lane bounds, exec-masked and
//  without the four-rows-per-trip form -- the lanes of a split batch then cost ~1.6x a full batch's per row.)
// A shape of fewer than five cells repeats its cell 0 in the slots behind: the fit is not changed by testing a cell twice,
// and the anchor hits of cell 0, counted apart, come off the total once per repeat at the end (the slots used to read a
// row of {all ones, 0} in a region of its own, which does not exist in a row-major table).
template <bool ANY_ONLY>
__device__ __forceinline__ uint32_t blk_shape_count(const WaveLds &L, const DistinctRegs &e, const bool active,
                                                    const int ya, const int n)
{
    // per cell: its table entry at origin row ya (row offset dy + 4, column offset dx + 4): the five addresses come
    // ready-made out of the table; a row further on is 72 bytes further on
    const char *base = reinterpret_cast<const char *>(&L) + ya * BLK_SH_ROW_BYTES;
    const char *cellrow[5] = {base + (e.o01 & 0xffffu), base + (e.o01 >> 16), base + (e.o23 & 0xffffu), base + (e.o23 >> 16),
                              base + (e.o4mp & 0xffffu)};
    uint32_t cnt = 0, cnt0 = 0;                          // anchor hits of the slots 1..4 / of cell 0
    auto one_row = [&](const int y) {
        uint2 v[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // the read stays inside the wave's pre-shifted table (rows 28..30 are there for the lanes that overshoot)
            CRL_BOUNDS_IN(cellrow[j] + y * BLK_SH_ROW_BYTES - reinterpret_cast<const char *>(&L), offsetof(WaveLds, u),
                          offsetof(WaveLds, u) + sizeof(L.u) - sizeof(uint2) + 1, 315);
            v[j] = *reinterpret_cast<const uint2 *>(cellrow[j] + y * BLK_SH_ROW_BYTES);
        }
        const uint32_t F = v[0].x & v[1].x & v[2].x & v[3].x & v[4].x;   // bit x+4: the shape fits at origin (x, y)
        // bit x+4 of v[j].y: cell j of the shape at origin (x, y) is an anchor.  (v_bcnt_u32_b32 adds its second operand:
        // accumulate in the instruction itself; left to the compiler the five counts go through a tree of v_add3)
        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt0) : "v"(F & v[0].y));
#pragma unroll
        for (int j = 1; j < 5; ++j) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt) : "v"(F & v[j].y));
    };
#ifndef BLK_COUNT_UNROLL
#define BLK_COUNT_UNROLL 4
#endif
    if (ANY_ONLY) {
#pragma nounroll
        for (int y = 0; y < n; ++y) {
            one_row(y);
            if (__ballot(active && (cnt | cnt0) != 0u)) break;     // (any hit is a hit of a real cell)
        }
    } else {
        // BLK_COUNT_UNROLL rows per trip, one after the other through the same registers: the row offset is an immediate of
        // the LDS reads, so the five table pointers move once per trip instead of once per row (5 of a row's 17 VALU) and the
        // loop control is shared.  (Two register sets with the next row's reads in flight behind the current row's
        // counting -- a software pipeline -- were 7 % SLOWER: 64 VGPRs, and the eight waves of a SIMD hide the latency.)
        int y = 0;
#pragma nounroll
        for (; y + BLK_COUNT_UNROLL - 1 < n; y += BLK_COUNT_UNROLL) {
#pragma unroll
            for (int u = 0; u < BLK_COUNT_UNROLL; ++u) one_row(y + u);
        }
#pragma nounroll
        for (; y < n; ++y) one_row(y);
    }
    // slots 1..4 hold cells - 1 real cells and 5 - cells repeats of cell 0
    const int cells = (int)((e.o4mp >> 20) & 7u) + 1;
    const uint32_t total = cnt + cnt0 * (uint32_t)(cells - 4);
    return active ? total : 0u;
}

// rows that can hold the origin of a shape touching an anchor of player q (wave-uniform)
__device__ __forceinline__ void blk_row_range(const WaveLds &L, const int q, const int lane, int &y0, int &y1)
{
    const bool has = lane < BN && L.ac[q][lane + 4].y != 0u;
    const unsigned long long m = __ballot(has);
    if (m == 0) { y0 = 0; y1 = -1; return; }
    const int lo = __builtin_ctzll(m), hi = 63 - __builtin_clzll(m);
    y0 = lo - 4 < 0 ? 0 : lo - 4;
    y1 = hi + 4 > BN - 1 ? BN - 1 : hi + 4;
}

// work list of the distinct oriented shapes of the pieces in `inv` (piece-major); returns its length.  One lane per distinct
// shape (91: lanes 0..63, then 0..26): a shape whose piece is held finds its place in the list by counting the held shapes
// before it -- two ballots and v_mbcnt, no scan -- and stores its index there, one byte store per lane.  (Round 2 / early
// round 3: one lane per PIECE, a wave scan of the pieces' shape counts and a loop of up to eight byte stores per lane; with
// the wave barrier behind it, 17 % of a ply together with the pre-shifted table.)  No barrier here: blk_build_shifted,
// which every caller runs next, ends with one.
// pa / pb: the pieces of the lane's two distinct shapes (blk_shape_owners) -- constants of the launch, which the rollout keeps
// in registers instead of reading them out of the tables on every ply
struct BlkOwners { uint32_t pa, pb; };
__device__ __forceinline__ BlkOwners blk_shape_owners(const BlkTables &T, const int lane)
{
    return BlkOwners{T.distinct[lane].piece, T.distinct[lane < NDISTINCT - 64 ? 64 + lane : 0].piece};
}

__device__ __forceinline__ int blk_build_items(WaveLds &L, const uint32_t inv, const int lane, const BlkOwners own)
{
    const uint32_t pa = own.pa, pb = own.pb;
    const bool ha = (inv >> pa) & 1u, hb = lane < NDISTINCT - 64 && ((inv >> pb) & 1u);
    const unsigned long long ma = __ballot(ha), mb = __ballot(hb);
    const uint32_t na = (uint32_t)__builtin_popcountll(ma);
    const uint32_t ra = __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
    const uint32_t rb = __builtin_amdgcn_mbcnt_lo((uint32_t)mb, na);          // (lanes 0..26 only: the low word)
    if (ha) L.items[ra] = (uint8_t)lane;                                        // index into T.distinct
    if (hb) L.items[rb] = (uint8_t)(64 + lane);
    return (int)(na + (uint32_t)__builtin_popcountll(mb));
}

// does player q have any legal action with inventory inv? (board.py:170-193 non-empty)
__device__ __forceinline__ bool blk_exists(const BlkTables &T, WaveLds &L, const int q, const uint32_t inv, const int lane,
                                           const BlkOwners *own = nullptr)
{
    int y0, y1;
    blk_row_range(L, q, lane, y0, y1);
    if (y1 < y0 || inv == 0) return false;
    if (inv & 1u) return true;                          // the monomino fits on any anchor (anchors are allowed cells)
    const int items = blk_build_items(L, inv, lane, own ? *own : blk_shape_owners(T, lane));
    blk_build_shifted(L, q, lane);
    for (int base = 0; base < items; base += 64) {
        const int i = base + lane;
        const bool active = i < items;
        CRL_BOUNDS_LT(active ? i : 0, NSHAPE, 311);
        CRL_BOUNDS_LT(active ? (int)L.items[i] : 0, NDISTINCT, 312);
        const DistinctRegs e = blk_load_distinct(T, active ? (int)L.items[i] : 0);
        const uint32_t c = blk_shape_count<true>(L, e, active, y0, y1 - y0 + 1);
        if (__ballot(c > 0)) return true;
    }
    return false;
}

// legal-action count per piece of player q into L.pcnt[], returns the total (valid_actions length)
// *piece_incl (lane p: the inclusive prefix of the per-piece counts up to piece p) is what level 1 of blk_select needs
#ifdef BLK_STAMPS
#define BLK_COUNT_STAMP(slot) do { if (stamp_acc_p) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc_p[slot] += now_ - *stamp_prev_p; *stamp_prev_p = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define BLK_COUNT_STAMP(slot) do { } while (0)
#endif
// the shape x row batches of a count pass and the per-piece totals; the work list (`items` entries), the pre-shifted table
// of player q, the cleared pcnt[] and the wave barrier behind them are the caller's (blk_count, or blk_prologue in the rollout).
// lo / hi: the first and the last board row that holds an anchor of the player.
// Every shape has its OWN origin rows: a cell with row offset dy can only touch an anchor when the origin lies in
// lo - dy .. hi - dy, so the shape's rows are lo - (its greatest dy) .. hi - (its least dy), cut to the board -- the anchor
// rows' span plus the shape's extent in rows (0..4), where one range for all shapes (round 2 / early round 3) was the span
// plus 8 cut to the board: 13.7 rows on average over random games for a span of 9.2.  A lane starts at its shape's first row
// and all lanes run the number of rows the widest shape of the batch needs (the work list is ordered by extent, so the
// shapes of a batch are alike and the widest is the last); rows behind a shape's own count nothing (blk_shape_count).
__device__ __forceinline__ uint32_t blk_count_batches(const BlkTables &T, WaveLds &L, const int lane, const int items, const int lo, const int hi,
                                                      uint32_t *piece_incl, uint32_t *piece_cnt
#ifdef BLK_STAMPS
                                                      , unsigned long long *stamp_acc_p = nullptr, unsigned long long *stamp_prev_p = nullptr
#endif
#ifdef BLK_COUNTERS
                                                      , unsigned long long *cnt_acc = nullptr
#endif
                                                      )
{
    // 64 lanes per batch of shapes.  A batch with at most 48 (16) shapes left gives each of up to 32 (16) shapes two (four) lanes, each with
    // its share of the origin rows -- the counts meet in pcnt[] anyway: the second batch of an early-game inventory (91
    // shapes: 64 + 27) and the only batch of a late one then take half or a quarter of the row loop.
    for (int base = 0; base < items; ) {
        const int left = items - base;
        // (33..48 left: 32 of them at two lanes each, the rest at four in the next batch -- three quarters of a full batch's rows)
        const int split_log = left <= 16 ? 2 : (left <= 48 ? 1 : 0);         // wave-uniform
        const int i = base + (lane >> split_log), part = lane & ((1 << split_log) - 1);
        const bool active = i < items;
        CRL_BOUNDS_LT(active ? i : 0, NSHAPE, 313);
        CRL_BOUNDS_LT(active ? (int)L.items[i] : 0, NDISTINCT, 314);
        const DistinctRegs e = blk_load_distinct(T, active ? (int)L.items[i] : 0);
        // the batch runs as many rows as its widest shape can need: the anchor rows' span plus that shape's extent, at most
        // the whole board -- the work list is ordered by extent, so the widest shape of a batch is its last (one v_readlane;
        // the exact maximum over the lanes' own counts, board edges included, costs a wave reduction per batch and buys nothing)
        const int rmax = (int)((e.rows >> 4) & 15u);
        const int first = max(lo + 4 - rmax, 0);
        const int last_lane = (min(left, 64 >> split_log) - 1) << split_log;
        const uint32_t rows_last = (uint32_t)__builtin_amdgcn_readlane((int)e.rows, last_lane);
        const int n_max = min(hi - lo + 1 + (int)((rows_last >> 4) & 15u) - (int)(rows_last & 15u), BN);
        const int share = (n_max + (1 << split_log) - 1) >> split_log;       // rows per lane, rounded up: the last part
        const int ya = first + part * share;                                  // overshoots by up to three rows
#ifdef BLK_COUNTERS
        if (cnt_acc) cnt_acc[3] += share;
#endif
        const uint32_t c = blk_shape_count<false>(L, e, active, ya, share) * ((e.o4mp >> 16) & 0xfu);
        if (c) atomicAdd(&L.pcnt[e.o4mp >> 24], c);
        base += 64 >> split_log;
    }
    BLK_COUNT_STAMP(6);                                   // ... the shape x row batches
    wave_sync();
    const uint32_t mine = lane < 32 ? L.pcnt[lane] : 0u;
    const uint32_t total = wave_scan_incl(mine, lane);
    if (piece_incl) *piece_incl = total;
    if (piece_cnt) *piece_cnt = mine;
    return (uint32_t)__builtin_amdgcn_readlane((int)total, 63);
}

__device__ __forceinline__ uint32_t blk_count(const BlkTables &T, WaveLds &L, const int q, const uint32_t inv, const int lane,
                                              uint32_t *piece_incl = nullptr, uint32_t *piece_cnt = nullptr)
{
    const unsigned long long m = __ballot(lane < BN && L.ac[q][lane + 4].y != 0u);   // rows with anchors (as blk_row_range)
    if (lane < 32) L.pcnt[lane] = 0;
    if (m == 0) { wave_sync(); if (piece_incl) *piece_incl = 0u; if (piece_cnt) *piece_cnt = 0u; return 0; }
    const int items = blk_build_items(L, inv, lane, blk_shape_owners(T, lane));
    blk_build_shifted(L, q, lane);                        // (its barrier also orders the pcnt clear and the work list)
    return blk_count_batches(T, L, lane, items, __builtin_ctzll(m), 63 - __builtin_clzll(m), piece_incl, piece_cnt);
}

// Everything a count pass of the ROLLOUT's mover needs, behind ONE wave barrier: player q's allowed / corner rows out of the
// board (as blk_prep), straight from the registers into the pre-shifted table as well (as blk_build_shifted, without
// reading the rows back), the rows that hold anchors as a ballot (as blk_row_range, without reading them back), the cleared
// pcnt[] and the work list.  As separate steps (prep, barrier, row range, work list, table, barrier) the same work sat in
// the dependent chain of every ply with three LDS round trips more.
struct BlkPrologue { int items; uint32_t anchor_rows; };        // anchor_rows: bit y = board row y holds an anchor of q
__device__ __forceinline__ BlkPrologue blk_prologue(WaveLds &L, const int lane, const int round, const int q, const uint32_t inv,
                                                    const BlkOwners own)
{
    // lanes 0..27 and 32..59: table row r = lane & 31 = board row y = r - 4 (rows outside the board are zero); the low half
    // writes shifts 0..4 and the rows themselves, the high half shifts 5..8
    const int r = lane & 31, y = r - 4;
    uint32_t allowed = 0u, corner = 0u;
    if (y >= 0 && y < BN) {
        const uint32_t any = L.occ[0][y] | L.occ[1][y] | L.occ[2][y] | L.occ[3][y];
        const uint32_t own = L.occ[q][y];
        const uint32_t up = y > 0 ? L.occ[q][y - 1] : 0u, dn = y < BN - 1 ? L.occ[q][y + 1] : 0u;
        const uint32_t orth = up | dn | (own << 1) | (own >> 1);           // computation.py:89-119
        allowed = ~any & ~orth & ROWMASK;                                 // computation.py:122-142
        if (round == 0) {                                                // board.py:177-179, corners of board.py:50
            const int cx = (q & 1) ? BN - 1 : 0, cy = (q & 2) ? BN - 1 : 0;
            corner = (y == cy) ? (allowed & (1u << cx)) : 0u;
        } else {                                                         // board.py:114-154
            const uint32_t ud = up | dn;
            corner = allowed & ((ud << 1) | (ud >> 1)) & ROWMASK;
        }
        if (lane < 32) L.ac[q][r] = make_uint2(allowed << 8, corner << 8);
    }
    const uint2 v = make_uint2(allowed << 8, corner << 8);
    const int s0 = lane < 32 ? 0 : 5;
    if (r < 28) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (s0 + k < 9) L.u.sh[r][s0 + k] = make_uint2(v.x >> (s0 + k), v.y >> (s0 + k));
    }
    BlkPrologue out;
    out.anchor_rows = ((uint32_t)__ballot(lane < 32 && corner != 0u)) >> 4;
    if (lane < 32) L.pcnt[lane] = 0;
    out.items = blk_build_items(L, inv, lane, own);
    wave_sync();
    return out;
}

struct BlkMove { int piece, x, y, orient, shift; };

// the r-th (0-based) legal action of player q in reference order; blk_count() must have filled L.pcnt
// HAVE_INCL: `piece_incl` is the scan blk_count handed out (the rollout keeps it in a register); else it is redone here
template <bool HAVE_INCL = false>
__device__ __forceinline__ BlkMove blk_select(const BlkTables &T, WaveLds &L, const int q, const uint32_t inv, uint32_t r, const int lane,
                                              const uint32_t piece_incl = 0u, const uint32_t piece_cnt = 0u, const uint32_t piece_cells = 0u)
{
    // level 1: the piece (pcnt is indexed by piece id; pieces not held count 0).  With HAVE_INCL everything comes out of
    // registers -- the scan and the per-piece counts as blk_count left them, the cell counts as loaded once per launch --
    // else out of LDS: each dependent LDS round trip of this serial pass is ~150 cycles of a ~6,000-cycle select.
    const uint32_t mine = HAVE_INCL ? piece_cnt : (lane < 32 ? L.pcnt[lane] : 0u);
    const uint32_t incl = HAVE_INCL ? piece_incl : wave_scan_incl(mine, lane);
    const unsigned long long hit = __ballot(r < incl);
    const int piece = __builtin_ctzll(hit);
    r -= (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), piece);
    const int n = HAVE_INCL ? __builtin_amdgcn_readlane((int)piece_cells, piece) : __builtin_amdgcn_readfirstlane((int)T.ncell[piece]);
    // the anchors: my row's (lanes 0..19), the rows that have any (a scalar mask), and the origin rows a shape touching
    // one can have (as blk_row_range)
    const uint32_t crow = lane < BN ? (L.ac[q][lane + 4].y >> 8) : 0u;
    uint32_t rows_mask = (uint32_t)__ballot(crow != 0u);
    // level 2: the anchors in row-major order, one lane per (orientation, shift) pair: lane o*n + j asks whether the piece
    // fits with its cell j on the anchor.  The ballot of those answers is the anchor's legal set in reference order, its
    // popcount the anchor's action count; both stay in scalar registers.  The anchors themselves are walked on the scalar
    // unit too -- rows with anchors from `rows_mask`, a row's anchors from its bits (v_readlane of the row lane's mask),
    // lowest set bit first: no anchor list in LDS (round 2 listed the anchors in LDS and paid a dependent LDS round trip
    // per anchor).
    // lane / n without a division: n is 1..5 and lane < 64 (floor(2^32 / n) + 1 is exact there)
    const uint32_t inv_n = n == 5 ? 858993460u : n == 4 ? 1073741825u : n == 3 ? 1431655766u : 2147483649u;
    const bool pair = lane < 8 * n;
    const int po = pair ? (n == 1 ? lane : (int)__umulhi((uint32_t)lane, inv_n)) : 0, pj = pair ? lane - po * n : 0;
    // NO fit table (round 2 / early round 3 built the fit masks of all 8 orientations x up to 20 origin rows in LDS on
    // every select: 41 % of the select's time, for a walk that visits two or three rows).  The pair lane fits its own
    // orientation with its own cell on the anchors of a row when the walk gets there: cell k of the shape then lies
    // (dy_k - dy_j) rows and (dx_k - dx_j) columns from the anchor, so
    //     legal anchors of row ay = AND_k  allowed[ay + dy_k - dy_j] >> (dx_k - dx_j)
    // -- five reads of the padded `ac` rows (rows outside the board are 0; the masks are stored << 8, which keeps every
    // shift a right shift), five shifts, two three-way ANDs per visited row, no table, no wave barrier.  Slots behind the
    // shape's cells repeat cell 0 (the same test twice).
    const uint2 cw = *reinterpret_cast<const uint2 *>(&T.cells[piece * 8 + po][0]);
    int rowoff[5];                                               // byte offset of cell k's row relative to ac[q][ay + 4 - 8]
    uint32_t colsh[5];                                           // right shift that brings cell k's column onto the anchor's
    {
        // all five cells at once, a byte each (round 3, second half; as ten chains of extract - subtract - scale this set-up was
        // ~35 vector instructions per ply): the row offsets dy_k + 8 - dy_j (0..16) times 8, the shifts 8 + dx_k - dx_j (0..16)
        uint32_t mineb = cw.x & 0xffu;
#pragma unroll
        for (int k = 1; k < 4; ++k) mineb = (pj == k) ? (cw.x >> (8 * k)) & 0xffu : mineb;
        mineb = (pj == 4) ? cw.y & 0xffu : mineb;
        const uint32_t dxj = mineb & 15u, dyj = mineb >> 4;
        const uint32_t ro4 = (((cw.x >> 4) & 0x0f0f0f0fu) + (8u - dyj) * 0x01010101u) << 3;
        const uint32_t cs4 = (cw.x & 0x0f0f0f0fu) + (8u - dxj) * 0x01010101u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rowoff[k] = (int)__builtin_amdgcn_ubfe(ro4, 8u * k, 8u);
            colsh[k] = __builtin_amdgcn_ubfe(cs4, 8u * k, 8u);
        }
        rowoff[4] = (int)((((cw.y >> 4) & 15u) + 8u - dyj) << 3);
        colsh[4] = (cw.y & 15u) + 8u - dxj;
    }
    const char *acq = reinterpret_cast<const char *>(&L.ac[q][4]) - 8 * (int)sizeof(uint2);   // (the row offsets carry a bias of 8 rows)
    // The walk starts from whichever end of the anchor order is nearer in rank: the piece's total is known (pcnt), so an
    // action in the upper half of the piece's range is counted down from the last anchor -- a quarter of the anchors per
    // select on average instead of half.
    const uint32_t ptotal = (uint32_t)__builtin_amdgcn_readlane((int)mine, piece);
    const bool back = r >= ((ptotal + 1u) >> 1);
    uint32_t rr = back ? ptotal - 1u - r : r;
    // A row at a time: every pair lane counts its legal anchors of the row (one AND, one popcount), a wave sum gives the
    // row's action count, and rows before the rank's are skipped whole; only inside the rank's row are the anchors looked
    // at one by one (a ballot, a popcount and a compare each on the scalar unit).  The two walking directions are two
    // instances of the loop (BACK a compile-time constant): selecting ctz / clz per row and per anchor cost a handful of
    // scalar instructions each time, and the scalar unit is what bounds this kernel.
    // Two plain loops with one exit each -- first the row, then the anchor in it (as nested loops with a `return` in the
    // middle the compiler built a maze of flag registers and branches: ~15 scalar instructions per anchor).  Both end by
    // themselves because r < total (the row counts add up to the piece's total, an anchor's counts to its row's); running
    // out of rows / anchors ends them too, so that an inconsistent state could give a wrong move but not a hang.
    auto walk = [&](auto back_tag) -> BlkMove {
        constexpr bool BACK = decltype(back_tag)::value;
        int ay;
        uint32_t cr, m, row_total;
        do {
            ay = BACK ? 31 - __builtin_clz(rows_mask) : __builtin_ctz(rows_mask);
            rows_mask &= ~(1u << ay);
            cr = (uint32_t)__builtin_amdgcn_readlane((int)crow, ay);              // this row's anchors, bit x
            // bit x: my (orientation, shift) pair is legal on anchor (x, ay)
            uint32_t fr = pair ? 0xffffffffu : 0u;
            const char *rowp = acq + ay * (int)sizeof(uint2);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                // the cell's padded row of player q: one of ac[q][0 .. 31]
                CRL_BOUNDS_IN(rowp + rowoff[k] - reinterpret_cast<const char *>(&L.ac[q][0]), 0, 32 * (int)sizeof(uint2) - (int)sizeof(uint2) + 1, 321);
                fr &= reinterpret_cast<const uint2 *>(rowp + rowoff[k])->x >> colsh[k];
            }
            m = fr & cr;
            row_total = (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl((uint32_t)__popc(m), lane), 63);
            if (rr < row_total) break;
            rr -= row_total;
        } while (rows_mask);
        int ax;
        uint32_t cnt;
        unsigned long long legal;
        bool mine_legal;
        do {                                                     // rr < row_total: the anchor is in this row
            ax = BACK ? 31 - __builtin_clz(cr) : __builtin_ctz(cr);
            const uint32_t bit = 1u << ax;
            cr &= ~bit;
            mine_legal = (m & bit) != 0u;
            legal = __ballot(mine_legal);
            cnt = (uint32_t)__builtin_popcountll(legal);
            if (rr < cnt) break;
            rr -= cnt;
        } while (cr);
        // level 3: the chosen legal pair at this anchor: the legal lane with exactly `want` legal lanes below it
        // (v_mbcnt: no scalar bit search)
        const uint32_t want = BACK ? cnt - 1u - rr : rr;
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(legal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)legal, 0u));
        const int lane_sel = __builtin_ctzll(__ballot(mine_legal && below == want) | (1ull << 63));
        BlkMove out = {piece, ax, ay, 0, 0};
        out.orient = n == 1 ? lane_sel : (int)__umulhi((uint32_t)lane_sel, inv_n);
        out.shift = lane_sel - out.orient * n;
        return out;
    };
    return back ? walk(std::true_type{}) : walk(std::false_type{});
}

// Board.update_board + AI.update_player as next_state runs them (BlokusEnvironment.py:417-420), for ANY action a caller
// hands over -- the reference does not test legality here (match_server.py:193 does, before the call):
//   * board.py:87-103 writes board_contents[y][x] = colour cell by cell under numpy's index rules: -20..-1 wrap around,
//     anything else outside 0..19 raises IndexError; a cell of another colour is overwritten;
//   * a shift id that names no cell of the piece raises IndexError (computation.py:218, offsets[offset_id]);
//   * ai.py:47 raises ValueError for a piece the mover does not hold -- after the board update, so the IndexError wins.
// next_state works on copies (:408-409): a raise leaves the state as it was.  Returns 0, CRL_BLOKUS_INDEX_ERROR or
// CRL_BLOKUS_VALUE_ERROR (wave-uniform); LDS rows, inventories and scores are touched only when it returns 0.
__device__ __forceinline__ int blk_apply(const BlkTables &T, WaveLds &L, const int q, const BlkMove &mv,
                                         uint32_t (&inv)[4], int (&score)[4], const int lane)
{
    // the cell bytes straight out of the table: lane j its own cell, everybody the anchored one
    const int n = __builtin_amdgcn_readfirstlane((int)T.ncell[mv.piece]);
    if (mv.shift >= n) return CRL_BLOKUS_INDEX_ERROR;
    const uint8_t *cells = &T.cells[mv.piece * 8 + mv.orient][0];
    const uint32_t oc = cells[mv.shift];
    const int ox = (int)(oc & 15u), oy = (int)(oc >> 4);
    const uint32_t cc = cells[lane < n ? lane : 0];
    int x = mv.x + (int)(cc & 15u) - ox, y = mv.y + (int)(cc >> 4) - oy;
    const bool mine = lane < n;
    if (__ballot(mine && (x < -BN || x >= BN || y < -BN || y >= BN)) != 0ull) return CRL_BLOKUS_INDEX_ERROR;
    uint32_t held = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) held = (c == q) ? inv[c] : held;
    if (!((held >> mv.piece) & 1u)) return CRL_BLOKUS_VALUE_ERROR;
    x += x < 0 ? BN : 0;                                   // numpy's negative indices
    y += y < 0 ? BN : 0;
    if (mine) {
        CRL_BOUNDS_IN(y * 32 + x, 0, BN * 32, 330);
        for (int c = 0; c < 4; ++c) {
            if (c == q) atomicOr(&L.occ[c][y], 1u << x);
            else atomicAnd(&L.occ[c][y], ~(1u << x));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c == q) {
            inv[c] &= ~(1u << mv.piece);
            score[c] += n + ((inv[c] == 0) ? (mv.piece == 0 ? 20 : 15) : 0);
        }
    }
    wave_sync();
    return 0;
}

// the board part of blk_apply for a move known to be legal (its cells are empty: only the mover's colour changes)
__device__ __forceinline__ void blk_place_legal(const BlkTables &T, WaveLds &L, const int q, const BlkMove &mv, const int lane, const int n)
{   // n: the piece's cell count (the rollout has it in a register)
    const uint8_t *cells = &T.cells[mv.piece * 8 + mv.orient][0];
    const uint32_t oc = cells[mv.shift];
    const int ox = (int)(oc & 15u), oy = (int)(oc >> 4);
    if (lane < n) {
        const uint32_t cc = cells[lane];
        const int x = mv.x + (int)(cc & 15u) - ox, y = mv.y + (int)(cc >> 4) - oy;
        if (x >= 0 && x < BN && y >= 0 && y < BN) atomicOr(&L.occ[q][y], 1u << x);
    }
    wave_sync();
}

struct BlkOutcome { int reward, terminal, winners; };

// BlokusEnvironment.py:424-447 once `any_move` is known
__device__ __forceinline__ BlkOutcome blk_outcome(const bool any_move, const int pl, const int (&score)[4])
{
    BlkOutcome o = {0, 0, 0};
    if (!any_move) {
        o.terminal = 1;
        int best = 0, mine = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { best = score[c] > best ? score[c] : best; mine = (c == pl) ? score[c] : mine; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o.winners |= (score[c] == best) << c;
            o.reward += (score[c] < mine || (score[c] == mine && c < pl)) ? 1 : 0;
        }
    }
    return o;
}

__device__ __forceinline__ void blk_load_state(WaveLds &L, const int64_t b, const int lane, const uint32_t *occ,
                                               const uint32_t *inv_g, const int32_t *score_g, uint32_t (&inv)[4], int (&score)[4])
{
    for (int i = lane; i < 4 * BN; i += 64) L.occ[i / BN][i % BN] = occ[b * 4 * BN + i];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                      // wave-uniform: keep them in SGPRs
        inv[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)inv_g[b * 4 + c]);
        score[c] = __builtin_amdgcn_readfirstlane(score_g[b * 4 + c]);
    }
    wave_sync();
}

__device__ __forceinline__ void blk_store_state(const WaveLds &L, const int64_t b, const int lane, uint32_t *occ,
                                                uint32_t *inv_g, int32_t *score_g, const uint32_t (&inv)[4], const int (&score)[4])
{
    wave_sync();
    for (int i = lane; i < 4 * BN; i += 64) occ[b * 4 * BN + i] = L.occ[i / BN][i % BN];
    if (lane < 4) {
        uint32_t iv = 0;
        int sc = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) { iv = (c == lane) ? inv[c] : iv; sc = (c == lane) ? score[c] : sc; }
        inv_g[b * 4 + lane] = iv;
        score_g[b * 4 + lane] = sc;
    }
}

__device__ __forceinline__ void blk_fresh(WaveLds &L, const int lane, uint32_t (&inv)[4], int (&score)[4])
{
    for (int i = lane; i < 4 * BN; i += 64) L.occ[i / BN][i % BN] = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) { inv[c] = (1u << NPIECE) - 1u; score[c] = 0; }
    wave_sync();
}

__device__ __forceinline__ BlkMove blk_decode(const int id)
{
    BlkMove mv;
    mv.shift = id % 5;
    mv.orient = (id / 5) & 7;
    const int cell = (id / 40) % 400;
    mv.piece = id / 16000;
    mv.x = cell % BN;
    mv.y = cell / BN;
    return mv;
}

// what the step entries accept: the dense ids above and, from CRL_BLOKUS_EXT_BASE on, ids whose index lies anywhere in
// [-20, 20) x [-20, 20) (next_state takes whatever string_to_action parsed, BlokusEnvironment.py:417); false: no such action
__device__ __forceinline__ bool blk_decode_any(const int id, BlkMove &mv)
{
    if (id < ACTION_IDS) { mv = blk_decode(id); return true; }
    const int a = id - CRL_BLOKUS_EXT_BASE;
    if (a >= CRL_BLOKUS_EXT_IDS) return false;
    mv.shift = a % 5;
    mv.orient = (a / 5) & 7;
    const int cell = (a / 40) % 1600;
    mv.piece = a / 64000;
    mv.x = cell % 40 - BN;
    mv.y = cell / 40 - BN;
    return true;
}

// the action part of next_state for an id >= 0: 0 or the code the reward slot carries
__device__ __forceinline__ int blk_play(const BlkTables &T, WaveLds &L, const int q, const int id,
                                        uint32_t (&inv)[4], int (&score)[4], const int lane)
{
    BlkMove mv;
    if (!blk_decode_any(id, mv)) return CRL_BLOKUS_BAD_ACTION;
    return blk_apply(T, L, q, mv, inv, score, lane);
}

#define BLK_SHARED_SETUP()                                                                        \
    __shared__ BlkTables T;                                                                       \
    __shared__ WaveLds Lw[4];                                                                     \
    for (int i = threadIdx.x; i < (int)(sizeof(BlkTables) / 4); i += blockDim.x)                  \
        reinterpret_cast<uint32_t *>(&T)[i] = reinterpret_cast<const uint32_t *>(tables)[i];      \
    __syncthreads();                                                                              \
    const int lane = threadIdx.x & 63;                                                            \
    const int wave_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* wave-uniform: SGPR addressing */ \
    WaveLds &L = Lw[wave_];                                                                       \
    /* rows -4..-1 and 20..27 of the padded ac[] rows are zero for the whole launch (only 0..19 are rewritten) */     \
    for (int i = lane; i < 4 * 32; i += 64) L.ac[i >> 5][i & 31] = make_uint2(0u, 0u);            \
    /* ... and so are rows 28..30 of the pre-shifted table (the count passes rewrite rows 0..27 only) */          \
    for (int i = lane; i < 3 * 9; i += 64) L.u.sh[28 + i / 9][i % 9] = make_uint2(0u, 0u);        \
    const int64_t b = (int64_t)blockIdx.x * 4 + wave_;                                            \
    if (b >= B) return;

// ---- diagnostic build only (-DBLK_STAMPS): where a rollout step spends its cycles.  Stamp values leave the
// kernel through g_blk_stamps alone; no output depends on them.  The shipped build compiles none of this.
#if defined(BLK_STAMPS) || defined(BLK_COUNTERS)
__device__ unsigned long long g_blk_stamps[8];
#endif
#ifdef BLK_STAMPS
#define BLK_STAMP(slot)                                                                      \
    do {                                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
        stamp_acc[slot] += now_ - stamp_prev;                                                \
        stamp_prev = __builtin_amdgcn_s_memtime();                                           \
    } while (0)
#else
#define BLK_STAMP(slot) do { } while (0)
#endif

// ---- kernels (one wave per game, 4 games per workgroup) ---------------------------------------------

__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_step_kernel(const BlkTables *__restrict__ tables, const int64_t B, uint32_t *__restrict__ occ,
                   uint32_t *__restrict__ inv_g, int32_t *__restrict__ score_g, int32_t *__restrict__ round_g,
                   int32_t *__restrict__ to_move_g, const int32_t *__restrict__ action, int8_t *__restrict__ reward,
                   uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners, const uint32_t flags)
{
    BLK_SHARED_SETUP();
    uint32_t inv[4];
    int score[4];
    blk_load_state(L, b, lane, occ, inv_g, score_g, inv, score);
    int round = __builtin_amdgcn_readfirstlane(round_g[b]), pl = __builtin_amdgcn_readfirstlane(to_move_g[b]) & 3;
    blk_prep(L, lane, round);                                    // allowed / corner rows of the PRE-move board (:424)
    const int id = __builtin_amdgcn_readfirstlane(action[b]);
    const int status = id >= 0 ? blk_play(T, L, pl, id, inv, score, lane) : 0;   // '' (pass) below 0 (:418)
    BlkOutcome out = {status, 0, 0};                             // the reference raises: the state stays, the code goes out
    if (status == 0) {
        bool any_move = false;
        for (int q = 0; q < 4 && !any_move; ++q) {               // old board, old round, NEW inventories (:424)
            uint32_t iq = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) iq = (c == q) ? inv[c] : iq;
            any_move = blk_exists(T, L, q, iq, lane);
        }
        out = blk_outcome(any_move, pl, score);
        round += (pl == 3) ? 1 : 0;                              // :446-447
        pl = (pl + 1) & 3;
    }
    if (lane == 0) {
        reward[b] = (int8_t)out.reward;
        terminal[b] = (uint8_t)out.terminal;
        winners[b] = (uint8_t)out.winners;
    }
    if (out.terminal && (flags & CRL_STEP_AUTO_RESET)) {
        blk_fresh(L, lane, inv, score);
        round = 0;
        pl = 0;
    }
    blk_store_state(L, b, lane, occ, inv_g, score_g, inv, score);
    if (lane == 0) { round_g[b] = round; to_move_g[b] = pl; }
}

// valid_actions: count, and optionally the dense id bitmap (bit id set = action id is legal)
__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_valid_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t *__restrict__ occ,
                    const uint32_t *__restrict__ inv_g, const int32_t *__restrict__ score_g,
                    const int32_t *__restrict__ round_g, const int32_t *__restrict__ to_move_g,
                    const int8_t *__restrict__ player, int32_t *__restrict__ count, uint32_t *__restrict__ mask)
{
    BLK_SHARED_SETUP();
    uint32_t inv[4];
    int score[4];
    blk_load_state(L, b, lane, occ, inv_g, score_g, inv, score);
    const int q = __builtin_amdgcn_readfirstlane(player ? (int)player[b] : to_move_g[b]) & 3;
    blk_prep(L, lane, __builtin_amdgcn_readfirstlane(round_g[b]));
    uint32_t iq = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) iq = (c == q) ? inv[c] : iq;
    const uint32_t total = blk_count(T, L, q, iq, lane);
    if (lane == 0 && count) count[b] = (int32_t)total;
    if (mask && total) {
        uint32_t *m = mask + b * (int64_t)MASK_WORDS;
        const int items = __popc(iq) * 8;
        for (int i = lane; i < items; i += 64) {
            const int piece = nth_set_bit32(iq, i >> 3), o = i & 7;
            const ShapeRegs s = blk_load_shape(T, piece, o);
            for (int y = 0; y < BN; ++y) {
                uint32_t F = 0xffffffffu, ct[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint2 ac = L.ac[q][y + s.ro(j)];
                    F &= ac.x >> s.sh(j);
                    ct[j] = ac.y >> s.sh(j);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    uint32_t hits = j < s.n ? (F & ct[j]) >> 4 : 0u;    // origin columns x with cell j on an anchor
                    while (hits) {
                        const int x = __builtin_ctz(hits);
                        hits &= hits - 1;
                        const int ax = x + s.sh(j) - 4, ay = y + s.ro(j) - 4;     // the anchor
                        const int id = ((piece * 400 + ay * BN + ax) * 8 + o) * 5 + j;
                        atomicOr(&m[id >> 5], 1u << (id & 31));
                    }
                }
            }
        }
    }
}

// the rollout's random agent for one step: the r-th legal action of the player to move, as a dense id
__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_sample_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi,
                     const uint64_t first_env_id, const uint32_t *__restrict__ occ, const uint32_t *__restrict__ inv_g,
                     const int32_t *__restrict__ score_g, const int32_t *__restrict__ round_g,
                     const int32_t *__restrict__ to_move_g, uint32_t *__restrict__ tcount, const int advance,
                     int32_t *__restrict__ action)
{
    BLK_SHARED_SETUP();
    uint32_t inv[4];
    int score[4];
    blk_load_state(L, b, lane, occ, inv_g, score_g, inv, score);
    const int pl = __builtin_amdgcn_readfirstlane(to_move_g[b]) & 3;
    blk_prep(L, lane, __builtin_amdgcn_readfirstlane(round_g[b]));
    uint32_t ip = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) ip = (c == pl) ? inv[c] : ip;
    const uint32_t total = blk_count(T, L, pl, ip, lane);
    const uint32_t tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)tcount[b]);
    const philox_out rnd = philox4x32_10((uint32_t)(first_env_id + (uint64_t)b), tc >> 2, 0u, CRL_TAG_BLOKUS, seed_lo, seed_hi);
    const uint32_t sel = tc & 3u;
    const uint32_t word = sel == 0 ? rnd.w[0] : sel == 1 ? rnd.w[1] : sel == 2 ? rnd.w[2] : rnd.w[3];
    int id = -1;
    if (total > 0) {
        const BlkMove mv = blk_select(T, L, pl, ip, __umulhi(word, total), lane);
        id = ((mv.piece * 400 + mv.y * BN + mv.x) * 8 + mv.orient) * 5 + mv.shift;
    }
    if (lane == 0) {
        action[b] = id;
        if (advance) tcount[b] = tc + 1u;
    }
}

// state_to_observation of game b for observer pl (BlokusEnvironment.py:752-768) by the wave that holds the game's row
// bitboards in LDS: board = -1 empty else (owner - observer) % 4, rotated by np.rot90(k=-pl), four cells (one dword) per
// lane and trip; pieces[r][i] = inventory bit i of player (r + observer) % 4; score rolled by -observer.
__device__ __forceinline__ void blk_write_observation(const uint32_t (&occ)[4][BN], const uint32_t (&inv)[4], const int (&score)[4],
                                                      const int pl, const int64_t b, const int lane, int8_t *__restrict__ obs_board,
                                                      uint8_t *__restrict__ obs_pieces, int32_t *__restrict__ obs_score)
{
    for (int d = lane; d < BN * BN / 4; d += 64) {
        const int i = d / (BN / 4), j0 = (d - i * (BN / 4)) * 4;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j0 + k;
            int y, x;                                            // source cell of np.rot90(m, k=-pl)[i][j]
            switch (pl) {
                case 0: y = i; x = j; break;
                case 1: y = BN - 1 - j; x = i; break;
                case 2: y = BN - 1 - i; x = BN - 1 - j; break;
                default: y = j; x = BN - 1 - i; break;
            }
            int v = -1;
#pragma unroll
            for (int c = 0; c < 4; ++c) v = ((occ[c][y] >> x) & 1u) ? ((c - pl) & 3) : v;
            word |= (uint32_t)(v & 0xff) << (8 * k);
        }
        reinterpret_cast<uint32_t *>(obs_board + b * (BN * BN))[d] = word;
    }
    for (int cell = lane; cell < 4 * NPIECE; cell += 64) {       // pieces[r][piece] of player (r + observer) % 4
        const int r = cell / NPIECE, piece = cell - r * NPIECE;
        uint32_t iv = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) iv = (c == ((r + pl) & 3)) ? inv[c] : iv;
        obs_pieces[b * 4 * NPIECE + cell] = (uint8_t)((iv >> piece) & 1u);
    }
    if (lane < 4) {                                              // np.roll(score, -observer)
        int sc = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) sc = (c == ((lane + pl) & 3)) ? score[c] : sc;
        obs_score[b * 4 + lane] = sc;
    }
}

// ---- fused per-ply call: [sample ->] next_state (auto-reset) -> len(valid_actions) and state_to_observation of the player
// to move next.  What BlokusVectorEnv / a learner runs every ply (BlokusEnvironment.py:357-451, :453-500, :721-768); as
// separate launches (sample, step, valid, observe) every one of them reloads the board into LDS and rebuilds the allowed /
// corner rows.  Here the wave that owns the game does all of it on its LDS copy.
__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_step_observe_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi,
                           const uint64_t first_env_id, uint32_t *__restrict__ occ, uint32_t *__restrict__ inv_g,
                           int32_t *__restrict__ score_g, int32_t *__restrict__ round_g, int32_t *__restrict__ to_move_g,
                           const int32_t *__restrict__ action, uint32_t *__restrict__ tcount, int8_t *__restrict__ reward,
                           uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners, int32_t *__restrict__ n_valid,
                           int8_t *__restrict__ obs_board, uint8_t *__restrict__ obs_pieces, int32_t *__restrict__ obs_score,
                           int8_t *__restrict__ obs_player, const uint32_t flags)
{
    BLK_SHARED_SETUP();
    uint32_t inv[4];
    int score[4];
    blk_load_state(L, b, lane, occ, inv_g, score_g, inv, score);
    int round = __builtin_amdgcn_readfirstlane(round_g[b]), pl = __builtin_amdgcn_readfirstlane(to_move_g[b]) & 3;
    blk_prep(L, lane, round);                                    // allowed / corner rows of the PRE-move board (:424)
    int id;
    if (action) {
        id = __builtin_amdgcn_readfirstlane(action[b]);
    } else {                                                     // the rollout's random agent at this game's step counter
        uint32_t ip = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) ip = (c == pl) ? inv[c] : ip;
        const uint32_t total = blk_count(T, L, pl, ip, lane);
        const uint32_t tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)tcount[b]);
        const philox_out rnd = philox4x32_10((uint32_t)(first_env_id + (uint64_t)b), tc >> 2, 0u, CRL_TAG_BLOKUS, seed_lo, seed_hi);
        const uint32_t sel = tc & 3u;
        const uint32_t word = sel == 0 ? rnd.w[0] : sel == 1 ? rnd.w[1] : sel == 2 ? rnd.w[2] : rnd.w[3];
        id = -1;
        if (total > 0) {
            const BlkMove mv = blk_select(T, L, pl, ip, __umulhi(word, total), lane);
            id = ((mv.piece * 400 + mv.y * BN + mv.x) * 8 + mv.orient) * 5 + mv.shift;
        }
        if (lane == 0) tcount[b] = tc + 1u;
    }
    const int status = id >= 0 ? blk_play(T, L, pl, id, inv, score, lane) : 0;   // '' (pass) below 0 (:418)
    BlkOutcome out = {status, 0, 0};                             // the reference raises: the state stays, the code goes out
    if (status == 0) {
        bool any_move = false;
        for (int q = 0; q < 4 && !any_move; ++q) {               // old board, old round, NEW inventories (:424)
            uint32_t iq = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) iq = (c == q) ? inv[c] : iq;
            any_move = blk_exists(T, L, q, iq, lane);
        }
        out = blk_outcome(any_move, pl, score);
        round += (pl == 3) ? 1 : 0;                              // :446-447
        pl = (pl + 1) & 3;
    }
    if (lane == 0) {
        reward[b] = (int8_t)out.reward;
        terminal[b] = (uint8_t)out.terminal;
        winners[b] = (uint8_t)out.winners;
    }
    if (out.terminal && (flags & CRL_STEP_AUTO_RESET)) {
        blk_fresh(L, lane, inv, score);
        round = 0;
        pl = 0;
    }
    blk_store_state(L, b, lane, occ, inv_g, score_g, inv, score);
    if (lane == 0) { round_g[b] = round; to_move_g[b] = pl; }
    // ---- what the next mover needs: its number of legal actions on the NEW board ...
    blk_prep(L, lane, round);
    uint32_t ip = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) ip = (c == pl) ? inv[c] : ip;
    const uint32_t total = blk_count(T, L, pl, ip, lane);
    if (lane == 0) { n_valid[b] = (int32_t)total; obs_player[b] = (int8_t)pl; }
    // ... and its observation (:752-768)
    blk_write_observation(L.occ, inv, score, pl, b, lane, obs_board, obs_pieces, obs_score);
}

__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_rollout_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi,
                      const uint64_t first_env_id, const int T_steps, uint32_t *__restrict__ occ, uint32_t *__restrict__ inv_g,
                      int32_t *__restrict__ score_g, int32_t *__restrict__ round_g, int32_t *__restrict__ to_move_g,
                      const crl_blokus_stats st)
{
    BLK_SHARED_SETUP();
    // Inventories and scores live in VECTOR registers, one player per lane (lane & 3; replicated over the wave): this
    // kernel is bound by the scalar unit (one instruction per ~4.2 cycles per SIMD, tools/ubench/valu_rate.hip), and as
    // four-entry arrays of scalar registers every "inventory of player pl" is a chain of three s_cselect and every update
    // eight of them (round 2 / early round 3: ~50 scalar instructions per ply in blk_apply alone).
    for (int i = lane; i < 4 * BN; i += 64) L.occ[i / BN][i % BN] = occ[b * 4 * BN + i];
    uint32_t vinv = inv_g[b * 4 + (lane & 3)];
    int vscore = score_g[b * 4 + (lane & 3)];
    wave_sync();
    int round = __builtin_amdgcn_readfirstlane(round_g[b]), pl = __builtin_amdgcn_readfirstlane(to_move_g[b]) & 3;
    uint32_t tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.tcount[b]);
    uint32_t ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)st.tstep[b]);
    const uint32_t g = (uint32_t)(first_env_id + (uint64_t)b);
    // bit q: player q is known to be out of moves for the rest of this game.  Once a player has no legal move in
    // a round >= 1 it can only pass, so its own cells and inventory stay fixed while `allowed` only shrinks:
    // the condition is permanent (round 0 is excluded: its single-corner anchor rule is not).  Purely a cache of
    // what the reference recomputes every step; it starts empty at kernel entry and at every reset.
    uint32_t dead = 0, can_move = 0;
    const uint32_t piece_cells = lane < 24 ? T.ncell[lane] : 0u;   // lane p: cells of piece p, for the whole launch
    const BlkOwners owners = blk_shape_owners(T, lane);            // lane l: the pieces of distinct shapes l and 64 + l
    uint32_t rnd_word = 0u;
#ifdef BLK_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#endif
#ifdef BLK_COUNTERS
    unsigned long long cnt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int t = 0; t < T_steps; ++t) {
        BLK_STAMP(0);
        const uint32_t ip = (uint32_t)__builtin_amdgcn_readlane((int)vinv, pl);
        // the mover's rows (another player's only when the game may end), pre-shifted table and work list, one barrier
        const BlkPrologue pro = blk_prologue(L, lane, round, pl, ip, owners);
        BLK_STAMP(1);
        uint32_t piece_incl = 0u, piece_cnt = 0u;
        uint32_t total = 0u;                                     // len(valid_actions) of the mover
        if (!((dead >> pl) & 1u) && pro.anchor_rows != 0u) {
            const int lo = __builtin_ctz(pro.anchor_rows), hi = 31 - __builtin_clz(pro.anchor_rows);
#ifdef BLK_COUNTERS   /* diagnostic builds only: how many row trips a count pass takes, against the (shape, row) pairs it covers */
            {
                const int y0_ = lo - 4 < 0 ? 0 : lo - 4, y1_ = hi + 4 > BN - 1 ? BN - 1 : hi + 4, n_rows_ = y1_ - y0_ + 1;
                cnt_acc[0] += 1; cnt_acc[1] += pro.items; cnt_acc[2] += n_rows_; cnt_acc[4] += hi - lo + 1;
                cnt_acc[5] += (pro.items * n_rows_ + 63) / 64; cnt_acc[6] += (pro.items * (hi - lo + 1 + 3) + 63) / 64;
                cnt_acc[7] += pro.items <= 32 ? 1 : 0;
            }
            total = blk_count_batches(T, L, lane, pro.items, lo, hi, &piece_incl, &piece_cnt, cnt_acc);
#elif defined(BLK_STAMPS)
            total = blk_count_batches(T, L, lane, pro.items, lo, hi, &piece_incl, &piece_cnt, stamp_acc, &stamp_prev);
#else
            total = blk_count_batches(T, L, lane, pro.items, lo, hi, &piece_incl, &piece_cnt);
#endif
        }
        if (total == 0 && round >= 1) dead |= 1u << pl;
        BLK_STAMP(2);
        // One Philox call serves 4 plies, and the calls of 16 plies are made TOGETHER on the vector unit: lanes 4k .. 4k+3
        // compute block (tc >> 4) * 4 + k and keep one of its words each, a ply takes its word out of lane tc & 15 with a
        // v_readlane.  (Round 2 ran
        // the ten rounds on the scalar unit, everything being wave-uniform: ~110 scalar instructions every fourth ply.
        // But this kernel is bound by the SCALAR unit -- it issues one instruction per ~4.2 cycles per SIMD whatever the
        // occupancy, half the rate of the vector unit (tools/ubench/valu_rate.hip, mix "salu"), and a ply has 408 scalar
        // against 529 vector instructions -- so uniform work is cheaper on the vector side.)
        if ((tc & 15u) == 0u || t == 0) {
            // lane j (j < 16) ends up with the word of ply (tc & ~15) + j: block j >> 2, word j & 3.  (The seed goes through
            // an opaque copy: the ten rounds' keys are then derived here, every 16 plies, instead of living in 20 scalar
            // registers across the whole step loop -- which had the compiler shuffle a dozen s_mov around the select walk.)
            uint32_t k0 = seed_lo, k1 = seed_hi;
            asm volatile("" : "+s"(k0), "+s"(k1));
            const philox_out r4 = philox4x32_10(g, ((tc >> 4) << 2) + (uint32_t)((lane >> 2) & 3), 0u, CRL_TAG_BLOKUS, k0, k1);
            const int ws = lane & 3;
            rnd_word = ws == 0 ? r4.w[0] : ws == 1 ? r4.w[1] : ws == 2 ? r4.w[2] : r4.w[3];
        }
        const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)rnd_word, (int)(tc & 15u));
        tc += 1;
        bool any_move = false;
        BlkMove mv = {0, 0, 0, 0, 0};
        if (total > 0) {
            const uint32_t r = __umulhi(word, total);
            mv = blk_select<true>(T, L, pl, ip, r, lane, piece_incl, piece_cnt, piece_cells);
            BLK_STAMP(3);
            // the mover keeps a move iff some OTHER piece of its inventory had one (new inventory, old board)
            any_move = total > (uint32_t)__builtin_amdgcn_readlane((int)piece_cnt, mv.piece);
        }
        // Does anybody else have a move?  The reference asks that of the PRE-move board (:424), so it is asked here before
        // the move is placed -- which lets the other players' rows be derived only now, when they are needed (late game).
        // `can_move` remembers who was found to have one on the board as it stands: while movers pass, the board and the
        // others' inventories stay what they were, so the answer does too (late in a game two or three players pass in a
        // row, and every one of those plies asked the same question of the same board).
        if (!any_move && (can_move & ~(1u << pl))) any_move = true;
        for (int q = 0; q < 4 && !any_move; ++q) {
            if (q == pl || ((dead >> q) & 1u)) continue;
            blk_prep(L, lane, round, q);
            const uint32_t iq = (uint32_t)__builtin_amdgcn_readlane((int)vinv, q);
            any_move = blk_exists(T, L, q, iq, lane, &owners);
            if (any_move) can_move |= 1u << q;
            if (!any_move && round >= 1) dead |= 1u << q;
        }
        BLK_STAMP(5);
        if (total > 0) {
            const int n = __builtin_amdgcn_readlane((int)piece_cells, mv.piece);
            blk_place_legal(T, L, pl, mv, lane, n);
            {                                                   // ai.py:44-54 for the mover's lane(s)
                const uint32_t left = vinv & ~(1u << mv.piece);
                const bool me = (lane & 3) == pl;
                vscore += me ? n + (left == 0u ? (mv.piece == 0 ? 20 : 15) : 0) : 0;
                vinv = me ? left : vinv;
            }
            can_move = 0;                                       // the board changed
            BLK_STAMP(4);
        }
        if (pl == 3) can_move = 0;                              // ... and so does the round (round 0 has its own anchor rule)
        const int mover = pl;
        round += (pl == 3) ? 1 : 0;
        pl = (pl + 1) & 3;
        ts += 1;
        if (!any_move) {                                        // terminal (BlokusEnvironment.py:424-440): the scores become scalars
            const int score[4] = {__builtin_amdgcn_readlane(vscore, 0), __builtin_amdgcn_readlane(vscore, 1),
                                  __builtin_amdgcn_readlane(vscore, 2), __builtin_amdgcn_readlane(vscore, 3)};
            const BlkOutcome out = blk_outcome(false, mover, score);
            dead = 0; can_move = 0;
            if (lane == 0) {        // episode statistics go straight to memory (this wave owns game b): nothing to carry
                st.n_episodes[b] += 1;
                st.len_sum[b] += ts;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    st.win_count[c * B + b] += (uint32_t)((out.winners >> c) & 1);
                    st.score_sum[c * B + b] += score[c];
                }
            }
            for (int i = lane; i < 4 * BN; i += 64) L.occ[i / BN][i % BN] = 0u;
            vinv = (1u << NPIECE) - 1u;
            vscore = 0;
            wave_sync();
            round = 0; pl = 0; ts = 0;
        }
    }
    BLK_STAMP(6);
#ifdef BLK_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_blk_stamps[i], stamp_acc[i]);
#endif
#ifdef BLK_COUNTERS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_blk_stamps[i], cnt_acc[i]);
#endif
    wave_sync();
    for (int i = lane; i < 4 * BN; i += 64) occ[b * 4 * BN + i] = L.occ[i / BN][i % BN];
    if (lane < 4) { inv_g[b * 4 + lane] = vinv; score_g[b * 4 + lane] = vscore; }
    if (lane == 0) {
        round_g[b] = round; to_move_g[b] = pl;
        st.tcount[b] = tc; st.tstep[b] = ts;
    }
    if (st.results && lane == 0) {     // packed result row for the gather, from the running totals: lane 0 wrote every one
        int32_t *row = st.results + b * 10;                          // of them itself (program order: no fence needed)
        row[0] = (int32_t)st.n_episodes[b];
        row[1] = (int32_t)st.len_sum[b];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            row[2 + c] = (int32_t)st.win_count[c * B + b];
            row[6 + c] = st.score_sum[c * B + b];
        }
    }
}

__global__ void __launch_bounds__(256)
blokus_reset_kernel(const int64_t B, const uint8_t *__restrict__ mask, uint32_t *__restrict__ occ, uint32_t *__restrict__ inv,
                    int32_t *__restrict__ score, int32_t *__restrict__ round, int32_t *__restrict__ to_move)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 4 * BN) return;
    const int64_t b = i / (4 * BN);
    if (mask && !mask[b]) return;
    occ[i] = 0u;
    const int k = (int)(i - b * 4 * BN);
    if (k < 4) { inv[b * 4 + k] = (1u << NPIECE) - 1u; score[b * 4 + k] = 0; }
    if (k == 0) { round[b] = 0; to_move[b] = 0; }
}

// board_contents export: int8 [B][20][20], 0 empty else colour (board.py:85), one thread per cell
__global__ void __launch_bounds__(256)
blokus_board_kernel(const int64_t B, const uint32_t *__restrict__ occ, int8_t *__restrict__ board)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * BN * BN) return;
    const int64_t b = i / (BN * BN);
    const int cell = (int)(i - b * BN * BN), y = cell / BN, x = cell - y * BN;
    int v = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) v = ((occ[(b * 4 + c) * BN + y] >> x) & 1u) ? c + 1 : v;
    board[i] = (int8_t)v;
}

// state_to_observation (BlokusEnvironment.py:721-768) for every game: one wave per game -- its 80 row words go through LDS
// (one coalesced load), then blk_write_observation emits the rotated board a dword per lane (the first version spent one
// thread, four cached row loads and a byte store on every cell)
__global__ void __launch_bounds__(256)
blokus_observe_kernel(const int64_t B, const uint32_t *__restrict__ occ, const uint32_t *__restrict__ inv,
                      const int32_t *__restrict__ score, const int8_t *__restrict__ player, int8_t *__restrict__ obs_board,
                      uint8_t *__restrict__ obs_pieces, int32_t *__restrict__ obs_score)
{
    __shared__ uint32_t s_occ[4][4][BN];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = (int64_t)blockIdx.x * 4 + wave;
    if (b >= B) return;
    for (int i = lane; i < 4 * BN; i += 64) s_occ[wave][i / BN][i % BN] = occ[b * 4 * BN + i];
    uint32_t iv[4];
    int sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        iv[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)inv[b * 4 + c]);
        sc[c] = __builtin_amdgcn_readfirstlane(score[b * 4 + c]);
    }
    const int pl = __builtin_amdgcn_readfirstlane((int)player[b]) & 3;
    wave_sync();
    blk_write_observation(s_occ[wave], iv, sc, pl, b, lane, obs_board, obs_pieces, obs_score);
}

// ---- the ordered legal-action list, compacted (what BlokusEnvironment.valid_actions returns, :453-500) -------------------
// count[b] and ids[b][0 .. min(count, cap)) = the dense ids of every legal action of `player` in ascending order, which is
// the reference's order (piece -> anchor row-major -> orientation -> shift, board.py:184-189).
//
// Round 4 formulation: WINDOW x PATTERN.  Whether (piece, orientation o, shift j) is legal on anchor a depends only on the
// 9 x 9 neighbourhood of a in the player's `allowed` mask: every cell of the oriented shape, moved so that its cell j lies
// on a, must be allowed, and no cell is further than 4 from another.  So
//   * once per call every anchor gets its window: 81 bits of NOT-allowed in three words (rows -4..-2, -1..1, 2..4 of nine
//     bits each), plus the anchor's part of the dense id -- 16 bytes per anchor in LDS (WaveLds::u.sh, free here);
//   * a table built at context creation holds, for every (piece, o, j), the 81-bit pattern of the cells relative to cell j
//     and the (piece, o, j) part of the id -- 16 bytes each, 13 KB, staged in LDS per workgroup;
//   * a candidate is legal iff (window & pattern) == 0: three v_and, one v_or3, one v_cmp.
// The candidates of a piece are enumerated in output order -- anchor, o, j -- 64 at a time: a unit of n passes (n = cells of
// the piece) covers 8 anchors x 8 orientations x n shifts, so a lane's (o, j) and its anchor slot are fixed per (piece, pass
// of the unit) and its pattern sits in registers for the whole piece.  A legal lane's place in the list is the running
// total plus the legal lanes below it (ballot + v_mbcnt), and ONE store per pass writes them: a raw buffer store whose
// descriptor ends at `cap` entries, so the hardware drops what does not fit (and everything when ids is NULL).
// (Round 3: lanes = (o, j), bit-parallel over the columns of an anchor row, then one ballot PER ANCHOR: 4,600 VALU + 2,900
// SALU per list, issue-bound at 157 us for 16,384 mid-game positions -- profiles/r4_step_api_*; the count pass it started
// with is gone too: the list counts itself.)
constexpr int BLK_WT = 128;                                  // anchors per chunk of the window table
constexpr int BLK_LIST_WAVES = 8;                            // waves (= games) per workgroup of the list kernel (4: 65.6 us, 8: 61.7, 16: 61.8)

// per-wave working set of the list kernel: 2.6 KB, so that 8 waves + the 13 KB pattern table stay below 35 KB per workgroup and
// four workgroups = 8 waves per SIMD fit a CU's 160 KB (with WaveLds and 4-wave groups it was 5 per SIMD)
struct ListLds {
    uint32_t occ[4][BN];     // board rows per colour
    uint2 acq[32];           // the player's {allowed << 8, corner << 8}, index y + 4; rows outside the board are 0
    uint4 wt[BLK_WT];        // window table: {~rows -4..-2, ~rows -1..1, ~rows 2..4, anchor part of the id}
};

// cells per piece, 3 bits each (a wave-uniform shift instead of a table read)
constexpr unsigned long long blk_ncell_packed()
{
    unsigned long long v = 0;
    for (int p = 0; p < NPIECE; ++p) v |= (unsigned long long)kPieceCells[p] << (3 * p);
    return v;
}

struct BlkPatterns { uint32_t e[NPIECE * 40][4]; };          // [piece * 40 + o * n + j] = {rows -4..-2, -1..1, 2..4, id part}
constexpr size_t BLK_PAT_OFFSET = (sizeof(BlkTables) + 15) & ~(size_t)15;

// false when a shape's cells do not fit a 9 x 9 window around each other (cannot happen with the 21 pieces)
bool build_patterns(const BlkTables &t, BlkPatterns &pt)
{
    memset(&pt, 0xff, sizeof(pt));                           // unused slots: bit 31 is set in every window word -> never legal
    for (int p = 0; p < NPIECE; ++p)
        for (int o = 0; o < 8; ++o) {
            const int n = t.ncell[p];
            for (int j = 0; j < n; ++j) {
                uint32_t w[3] = {0u, 0u, 0u};
                const int jx = t.cells[p * 8 + o][j] & 15, jy = t.cells[p * 8 + o][j] >> 4;
                for (int c = 0; c < n; ++c) {
                    const int dx = (t.cells[p * 8 + o][c] & 15) - jx, dy = (t.cells[p * 8 + o][c] >> 4) - jy;
                    if (dx < -4 || dx > 4 || dy < -4 || dy > 4) return false;
                    const int idx = (dy + 4) * 9 + (dx + 4);
                    w[idx / 27] |= 1u << (idx % 27);
                }
                uint32_t *e = pt.e[p * 40 + o * n + j];
                e[0] = w[0]; e[1] = w[1]; e[2] = w[2];
                e[3] = (uint32_t)(p * 16000 + o * 5 + j);
            }
        }
    return true;
}

// window table entries [0, round-up-to-8 of the chunk's anchors) for the anchors [lo, lo + BLK_WT) of the player
__device__ __forceinline__ void blk_list_windows(ListLds &S, const int lane, const uint32_t crow, const uint32_t row_start,
                                                 const int lo, const int n_c)
{
    uint32_t *Ww = reinterpret_cast<uint32_t *>(S.wt);
    // lane y < 20 hands out its row's anchors: entry k gets (x, y) in its fourth word for now
    uint32_t m = crow, idx = row_start;
    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
        if (m != 0u) {
            const int ax = __builtin_ctz(m);
            m &= m - 1u;
            const uint32_t k = idx - (uint32_t)lo;
            if (k < (uint32_t)BLK_WT) Ww[k * 4 + 3] = (uint32_t)ax | ((uint32_t)lane << 8);
            CRL_BOUNDS_LT(ax, BN, 301);
            idx += 1u;
        }
    }
    wave_sync();
    const int padded = (n_c + 7) & ~7;
#pragma unroll
    for (int h = 0; h < BLK_WT / 64; ++h) {
        const int k = h * 64 + lane;
        if (k < n_c) {
            const uint32_t xy = Ww[k * 4 + 3];
            const int ax = (int)(xy & 0xffu), ay = (int)(xy >> 8);
            uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
            for (int dy = 0; dy < 9; ++dy) {             // acq[y + 4] holds row y << 8: window column dx is bit ax + dx + 8
                CRL_BOUNDS_LT(ay + dy, 32, 302);                 // padded rows -4 .. 27
                CRL_BOUNDS_LT(ax + 4 + 9, 33, 303);              // the nine window columns lie inside the 32-bit row word
                const uint32_t row = S.acq[ay + dy].x;
                w[dy / 3] |= ((row >> (ax + 4)) & 0x1ffu) << (9 * (dy % 3));
            }
            S.wt[k] = make_uint4(~w[0], ~w[1], ~w[2], (uint32_t)((ay * BN + ax) * 40));
        } else if (k < padded) {
            S.wt[k] = make_uint4(~0u, ~0u, ~0u, 0u);     // the tail of the last unit of 8: never legal
        }
    }
    wave_sync();
}

// the units of one piece over one chunk of anchors, N = the piece's cells (compile time: N passes per unit, no loop control
// inside a unit).  Returns the new running total.
template <int N>
__device__ __forceinline__ uint32_t blk_list_units(const ListLds &S, const uint4 *__restrict__ Pp, const int lane, const int n_c,
                                                   const __amdgpu_buffer_rsrc_t out, uint32_t base)
{
    // pass s of a unit: lane -> candidate s * 64 + lane of the unit's 8 anchors x 8 N (o, j) pairs
    uint4 pat[N];
    int slot[N];                                                 // anchor of the unit (0..7)
    // (through an opaque copy of the lane id: otherwise the compiler computes the decode of all five N up front, keeps
    // ~15 values alive across the piece loop and spills at the 64 registers of 8 waves per SIMD: 134 us instead of 62)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int s = 0; s < N; ++s) {                                // (unsigned: a shift / one v_mul_hi per quotient; signed division
        const uint32_t item = (uint32_t)(s * 64) + (uint32_t)ln;  // by 32 alone was five instructions)
        const uint32_t a = item / (uint32_t)(8 * N);
        slot[s] = (int)a;
        CRL_BOUNDS_LT(a, 8u, 304);
        CRL_BOUNDS_LT(item - a * (uint32_t)(8 * N), 40u, 305);   // a piece's 8 N <= 40 patterns
        pat[s] = Pp[item - a * (uint32_t)(8 * N)];
    }
    for (int u = 0; u < n_c; u += 8) {
        uint4 w[N];                                              // the unit's window reads go out together
#pragma unroll
        for (int s = 0; s < N; ++s) {
            CRL_BOUNDS_LT(u + slot[s], BLK_WT, 306);             // inside the (padded) chunk of the window table
            w[s] = S.wt[u + slot[s]];
        }
#pragma unroll
        for (int s = 0; s < N; ++s) {
            const bool legal = ((w[s].x & pat[s].x) | (w[s].y & pat[s].y) | (w[s].z & pat[s].z)) == 0u;
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(legal);
            if (lm == 0ull) continue;                            // (measured: skip + masked store 61.7 us, no skip 63.3,
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
            // every lane storing, the illegal ones past the descriptor's end, 65.8)
            if (legal) __builtin_amdgcn_raw_buffer_store_b32(w[s].w + pat[s].w, out, (int)((base + below) << 2), 0, 0);
            base += (uint32_t)__builtin_popcountll(lm);
        }
    }
    return base;
}

__global__ void __launch_bounds__(64 * BLK_LIST_WAVES, 8)
blokus_list_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t *__restrict__ occ,
                   const uint32_t *__restrict__ inv_g, const int32_t *__restrict__ score_g,
                   const int32_t *__restrict__ round_g, const int32_t *__restrict__ to_move_g,
                   const int8_t *__restrict__ player, int32_t *__restrict__ ids, int32_t *__restrict__ count, const int cap)
{
    (void)score_g;
    __shared__ uint4 Pt[NPIECE * 40];
    __shared__ ListLds Sw[BLK_LIST_WAVES];
    {
        const uint4 *pat = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(tables) + BLK_PAT_OFFSET);
        for (int i = threadIdx.x; i < NPIECE * 40; i += blockDim.x) Pt[i] = pat[i];
    }
    const int lane = threadIdx.x & 63;
    const int wave_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    ListLds &S = Sw[wave_];
    if (lane < 32) S.acq[lane] = make_uint2(0u, 0u);             // the padding rows stay zero
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * BLK_LIST_WAVES + wave_;
    if (b >= B) return;
    for (int i = lane; i < 4 * BN; i += 64) S.occ[i / BN][i % BN] = occ[b * 4 * BN + i];
    const int q = __builtin_amdgcn_readfirstlane(player ? (int)player[b] : to_move_g[b]) & 3;
    const uint32_t iq = (uint32_t)__builtin_amdgcn_readfirstlane((int)inv_g[b * 4 + q]);
    const int round = __builtin_amdgcn_readfirstlane(round_g[b]);
    wave_sync();
    // the anchors in row-major order: lane y < 20 knows its row's and where they start in the list of anchors
    uint32_t crow = 0u;
    if (lane < BN) {
        const uint2 ac = blk_prep_row(S.occ, q, lane, round);
        S.acq[lane + 4] = ac;
        crow = ac.y >> 8;
    }
    const uint32_t crow_n = (uint32_t)__popc(crow);
    const uint32_t incl = wave_scan_incl(crow_n, lane);
    const int A = __builtin_amdgcn_readlane((int)incl, 63);
    uint32_t base = 0;                                           // actions listed so far (wave-uniform)
    if (A > 0 && iq != 0u) {
        const int nchunks = (A + BLK_WT - 1) / BLK_WT;           // 1 unless a player has more than 128 anchors
        if (nchunks == 1) blk_list_windows(S, lane, crow, incl - crow_n, 0, A);          // (its first barrier covers acq)
        // stores go through a buffer descriptor that ends after `cap` ids: what does not fit is dropped by the hardware
        const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(
            ids ? (void *)(ids + b * (int64_t)cap) : (void *)count, 0, ids ? cap * (int)sizeof(int32_t) : 0, 0x00020000);
        for (int piece = 0; piece < NPIECE; ++piece) {
            if (!((iq >> piece) & 1u)) continue;
            const int n = (int)((blk_ncell_packed() >> (3 * piece)) & 7ull);
            const uint4 *Pp = &Pt[piece * 40];
            for (int c = 0; c < nchunks; ++c) {
                const int n_c = A - c * BLK_WT < BLK_WT ? A - c * BLK_WT : BLK_WT;
                if (nchunks > 1) blk_list_windows(S, lane, crow, incl - crow_n, c * BLK_WT, n_c);
                switch (n) {
                    case 1: base = blk_list_units<1>(S, Pp, lane, n_c, out, base); break;
                    case 2: base = blk_list_units<2>(S, Pp, lane, n_c, out, base); break;
                    case 3: base = blk_list_units<3>(S, Pp, lane, n_c, out, base); break;
                    case 4: base = blk_list_units<4>(S, Pp, lane, n_c, out, base); break;
                    default: base = blk_list_units<5>(S, Pp, lane, n_c, out, base); break;
                }
            }
        }
    }
    if (lane == 0 && count) count[b] = (int32_t)base;
}

// the rank[b]-th legal action of `player` in reference order as a dense id (-1 when rank is outside [0, count)):
// blk_select with the rank from memory instead of the rollout's Philox draw
__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_select_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t *__restrict__ occ,
                     const uint32_t *__restrict__ inv_g, const int32_t *__restrict__ score_g,
                     const int32_t *__restrict__ round_g, const int32_t *__restrict__ to_move_g,
                     const int8_t *__restrict__ player, const int32_t *__restrict__ rank, int32_t *__restrict__ action,
                     int32_t *__restrict__ count)
{
    BLK_SHARED_SETUP();
    uint32_t inv[4];
    int score[4];
    blk_load_state(L, b, lane, occ, inv_g, score_g, inv, score);
    const int q = __builtin_amdgcn_readfirstlane(player ? (int)player[b] : to_move_g[b]) & 3;
    blk_prep(L, lane, __builtin_amdgcn_readfirstlane(round_g[b]));
    uint32_t iq = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) iq = (c == q) ? inv[c] : iq;
    const uint32_t total = blk_count(T, L, q, iq, lane);
    const int r = __builtin_amdgcn_readfirstlane(rank[b]);
    int id = -1;
    if (r >= 0 && (uint32_t)r < total) {
        const BlkMove mv = blk_select(T, L, q, iq, (uint32_t)r, lane);
        id = ((mv.piece * 400 + mv.y * BN + mv.x) * 8 + mv.orient) * 5 + mv.shift;
    }
    if (lane == 0) {
        action[b] = id;
        if (count) count[b] = (int32_t)total;
    }
}

// is_valid_action (BlokusEnvironment.py:667-719: membership in valid_actions) without the enumeration: the piece is held,
// `shift` names one of its cells, that cell's target is an anchor of the player, and every cell lands on an allowed cell.
// FIT_ONLY: only the last condition (and the shift naming a cell) -- what Board.check_orientation_shifts asks of ONE shift
// (board.py:156-168, computation.py:144-180 check_shifted): the index need not be an anchor, the piece need not be held;
// inv_g / round_g / to_move_g are not read then.
template <bool FIT_ONLY>
__global__ void __launch_bounds__(256, BLK_WAVES_PER_SIMD)
blokus_is_valid_kernel(const BlkTables *__restrict__ tables, const int64_t B, const uint32_t *__restrict__ occ,
                       const uint32_t *__restrict__ inv_g, const int32_t *__restrict__ round_g, const int32_t *__restrict__ to_move_g,
                       const int8_t *__restrict__ player, const int32_t *__restrict__ action, uint8_t *__restrict__ ok)
{
    BLK_SHARED_SETUP();
    for (int i = lane; i < 4 * BN; i += 64) L.occ[i / BN][i % BN] = occ[b * 4 * BN + i];
    wave_sync();
    const int q = __builtin_amdgcn_readfirstlane(player ? (int)player[b] : to_move_g[b]) & 3;
    blk_prep(L, lane, FIT_ONLY ? 1 : __builtin_amdgcn_readfirstlane(round_g[b]), q);
    const uint32_t iq = FIT_ONLY ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)inv_g[b * 4 + q]);
    const int id = __builtin_amdgcn_readfirstlane(action[b]);
    bool good = false;
    if (id >= 0 && id < ACTION_IDS) {
        const BlkMove mv = blk_decode(id);
        const ShapeRegs s = blk_load_shape(T, mv.piece, mv.orient);
        if ((FIT_ONLY || ((iq >> mv.piece) & 1u)) && mv.shift < s.n) {
            int ox = 0, oy = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) { ox = (j == mv.shift) ? s.sh(j) : ox; oy = (j == mv.shift) ? s.ro(j) : oy; }
            bool cell_ok = true;
            if (lane < s.n) {
                int cx = 0, cy = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) { cx = (j == lane) ? s.sh(j) : cx; cy = (j == lane) ? s.ro(j) : cy; }
                const int x = mv.x + cx - ox, y = mv.y + cy - oy;
                cell_ok = x >= 0 && x < BN && y >= 0 && y < BN && ((L.ac[q][y + 4].x >> (x + 8)) & 1u);
            }
            const bool anchor = FIT_ONLY || ((L.ac[q][mv.y + 4].y >> (mv.x + 8)) & 1u);
            good = anchor && __ballot(!cell_ok) == 0ull;
        }
    }
    if (lane == 0) ok[b] = good ? 1 : 0;
}

// occ rows from Board.board_contents (int8 [B][20][20], 0 empty else colour): the inverse of blokus_board_kernel
__global__ void __launch_bounds__(256)
blokus_pack_kernel(const int64_t B, const int8_t *__restrict__ board, uint32_t *__restrict__ occ)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 4 * BN) return;
    const int64_t b = i / (4 * BN);
    const int k = (int)(i - b * 4 * BN), c = k / BN, y = k - c * BN;
    const int8_t *row = board + (b * BN + y) * BN;
    uint32_t m = 0;
    for (int x = 0; x < BN; ++x) m |= (row[x] == c + 1) ? (1u << x) : 0u;
    occ[i] = m;
}

} // namespace

void crl_blokus_free(void *tables)
{
    if (tables) (void)hipFree(tables);
}

#define BLK_CTX_CHECK(fn)                                                                       \
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_BLOKUS && ctx->blokus, fn ": ctx is not a blokus context"); \
    CRL_REQUIRE(B > 0 && B <= ((int64_t)1 << 28), fn ": B=%lld out of range", (long long)B)

int crl_blokus_bounds(unsigned int *out4)
{
    CRL_BOUNDS_READBACK(out4);
    return CRL_OK;
}

extern "C" {

int crl_blokus_create(crl_ctx **out)
{
    CRL_REQUIRE(out != nullptr, "crl_blokus_create: out is NULL");
    BlkTables host;
    build_tables(host);
    build_distinct(host);
    {
        int nd = 0;
        for (int p = 0; p < NPIECE; ++p) nd += host.nuniq[p];
        CRL_REQUIRE(nd == NDISTINCT, "crl_blokus_create: %d distinct oriented shapes, expected %d", nd, NDISTINCT);
    }
    // one allocation: the tables, then (16-byte aligned) the window patterns of the list pass
    struct PatternBuf {                                           // 13 KB: not on the stack, freed on every path
        BlkPatterns *p = new BlkPatterns;
        ~PatternBuf() { delete p; }
    } patterns;
    CRL_REQUIRE(build_patterns(host, *patterns.p), "crl_blokus_create: a piece does not fit the 9 x 9 window of the list pass");
    void *dev = nullptr;
    CRL_HIP(hipMalloc(&dev, BLK_PAT_OFFSET + sizeof(BlkPatterns)));
    hipError_t e = hipMemcpy(dev, &host, sizeof(BlkTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy((char *)dev + BLK_PAT_OFFSET, patterns.p, sizeof(BlkPatterns), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(dev);
        crl_set_error("crl_blokus_create: hipMemcpy failed: %s", hipGetErrorString(e));
        return CRL_EHIP;
    }
    crl_ctx *c = new crl_ctx();
    memset(c, 0, sizeof(*c));
    c->game = CRL_GAME_BLOKUS;
    c->blokus = dev;
    *out = c;
    return CRL_OK;
}

int crl_blokus_stamps(uint64_t *out8, int reset)
{
    CRL_REQUIRE(out8 != nullptr, "crl_blokus_stamps: out8 is NULL");
#if defined(BLK_STAMPS) || defined(BLK_COUNTERS)
    CRL_HIP(hipDeviceSynchronize());
    CRL_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_blk_stamps), 8 * sizeof(uint64_t)));
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        CRL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_blk_stamps), zero, sizeof(zero)));
    }
    return 1;
#else
    (void)reset;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    return 0;
#endif
}

int crl_blokus_placement(int piece, int orient, int shift, int8_t *cells_xy)
{
    CRL_REQUIRE(piece >= 0 && piece < NPIECE && orient >= 0 && orient < 8 && cells_xy, "crl_blokus_placement: bad argument");
    CRL_REQUIRE(shift >= 0 && shift < kPieceCells[piece], "crl_blokus_placement: shift %d out of range for piece %d", shift, piece);
    BlkTables t;
    build_tables(t);
    const uint8_t *c = t.cells[piece * 8 + orient];
    for (int j = 0; j < kPieceCells[piece]; ++j) {
        cells_xy[2 * j] = (int8_t)((c[j] & 15) - (c[shift] & 15));
        cells_xy[2 * j + 1] = (int8_t)((c[j] >> 4) - (c[shift] >> 4));
    }
    return kPieceCells[piece];
}

int crl_blokus_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask, uint32_t *occ, uint32_t *inv, int32_t *score,
                     int32_t *round, int32_t *to_move, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_reset");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_reset: NULL state pointer");
    const int64_t n = B * 4 * BN;
    hipLaunchKernelGGL(blokus_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, mask, occ, inv, score, round, to_move);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_step(const crl_ctx *ctx, int64_t B, uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                    const int32_t *action, int8_t *reward, uint8_t *terminal, uint8_t *winners, uint32_t flags, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_step");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_step: NULL state pointer");
    CRL_REQUIRE(action && reward && terminal && winners, "crl_blokus_step: NULL action/output pointer");
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "crl_blokus_step: unknown flags 0x%x", flags);
    hipLaunchKernelGGL(blokus_step_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, inv, score, round, to_move, action, reward, terminal, winners, flags);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                     const int32_t *round, const int32_t *to_move, const int8_t *player, int32_t *count, uint32_t *mask, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_valid");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_valid: NULL state pointer");
    CRL_REQUIRE(count || mask, "crl_blokus_valid: nothing to compute (count and mask are NULL)");
    if (mask) CRL_HIP(hipMemsetAsync(mask, 0, (size_t)B * MASK_WORDS * 4, (hipStream_t)stream));
    hipLaunchKernelGGL(blokus_valid_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, inv, score, round, to_move, player, count, mask);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_valid_list(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                          const int32_t *round, const int32_t *to_move, const int8_t *player, int32_t *ids, int32_t *count,
                          int cap, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_valid_list");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_valid_list: NULL state pointer");
    CRL_REQUIRE(ids || count, "crl_blokus_valid_list: nothing to compute (ids and count are NULL)");
    CRL_REQUIRE(ids == nullptr || (cap > 0 && cap <= (1 << 28)), "crl_blokus_valid_list: cap=%d out of range 1..2^28", cap);
    hipLaunchKernelGGL(blokus_list_kernel, dim3((unsigned)((B + BLK_LIST_WAVES - 1) / BLK_LIST_WAVES)), dim3(64 * BLK_LIST_WAVES), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, inv, score, round, to_move, player, ids, count, cap);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_select(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                      const int32_t *round, const int32_t *to_move, const int8_t *player, const int32_t *rank,
                      int32_t *action, int32_t *count, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_select");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_select: NULL state pointer");
    CRL_REQUIRE(rank && action, "crl_blokus_select: NULL rank / action pointer");
    hipLaunchKernelGGL(blokus_select_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, inv, score, round, to_move, player, rank, action, count);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_is_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                        const int32_t *round, const int32_t *to_move, const int8_t *player, const int32_t *action,
                        uint8_t *ok, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_is_valid");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_is_valid: NULL state pointer");
    CRL_REQUIRE(action && ok, "crl_blokus_is_valid: NULL action / ok pointer");
    (void)score;
    hipLaunchKernelGGL(blokus_is_valid_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, inv, round, to_move, player, action, ok);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_fits(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const int8_t *player, const int32_t *action, uint8_t *ok,
                    void *stream)
{
    BLK_CTX_CHECK("crl_blokus_fits");
    CRL_REQUIRE(occ && player && action && ok, "crl_blokus_fits: NULL pointer");
    hipLaunchKernelGGL(blokus_is_valid_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, occ, (const uint32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, player, action, ok);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_pack(const crl_ctx *ctx, int64_t B, const int8_t *board, uint32_t *occ, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_pack");
    CRL_REQUIRE(board && occ, "crl_blokus_pack: NULL pointer");
    const int64_t n = B * 4 * BN;
    hipLaunchKernelGGL(blokus_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, board, occ);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_board(const crl_ctx *ctx, int64_t B, const uint32_t *occ, int8_t *board, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_board");
    CRL_REQUIRE(occ && board, "crl_blokus_board: NULL pointer");
    const int64_t n = B * BN * BN;
    hipLaunchKernelGGL(blokus_board_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, occ, board);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_observe(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                       const int8_t *player, int8_t *obs_board, uint8_t *obs_pieces, int32_t *obs_score, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_observe");
    CRL_REQUIRE(occ && inv && score && player, "crl_blokus_observe: NULL input pointer");
    CRL_REQUIRE(obs_board && obs_pieces && obs_score, "crl_blokus_observe: NULL output pointer");
    CRL_REQUIRE((((uintptr_t)obs_board) & 3) == 0, "crl_blokus_observe: obs_board must be 4-byte aligned");
    hipLaunchKernelGGL(blokus_observe_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       B, occ, inv, score, player, obs_board, obs_pieces, obs_score);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, const uint32_t *occ,
                      const uint32_t *inv, const int32_t *score, const int32_t *round, const int32_t *to_move,
                      uint32_t *tcount, int advance, int32_t *action, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_sample");
    CRL_REQUIRE(occ && inv && score && round && to_move && tcount && action, "crl_blokus_sample: NULL pointer");
    hipLaunchKernelGGL(blokus_sample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id,
                       occ, inv, score, round, to_move, tcount, advance, action);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                            uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                            const int32_t *action, uint32_t *tcount, int8_t *reward, uint8_t *terminal, uint8_t *winners,
                            int32_t *n_valid, int8_t *obs_board, uint8_t *obs_pieces, int32_t *obs_score, int8_t *obs_player,
                            uint32_t flags, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_step_observe");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_step_observe: NULL state pointer");
    CRL_REQUIRE(action || tcount, "crl_blokus_step_observe: action and tcount are both NULL (nothing to play)");
    CRL_REQUIRE(reward && terminal && winners && n_valid, "crl_blokus_step_observe: NULL step output pointer");
    CRL_REQUIRE(obs_board && obs_pieces && obs_score && obs_player, "crl_blokus_step_observe: NULL observation pointer");
    CRL_REQUIRE((((uintptr_t)obs_board) & 3) == 0, "crl_blokus_step_observe: obs_board must be 4-byte aligned");
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "crl_blokus_step_observe: unknown flags 0x%x", flags);
    hipLaunchKernelGGL(blokus_step_observe_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id,
                       occ, inv, score, round, to_move, action, tcount, reward, terminal, winners, n_valid,
                       obs_board, obs_pieces, obs_score, obs_player, flags);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_blokus_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                       uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                       crl_blokus_stats st, void *stream)
{
    BLK_CTX_CHECK("crl_blokus_rollout");
    CRL_REQUIRE(occ && inv && score && round && to_move, "crl_blokus_rollout: NULL state pointer");
    CRL_REQUIRE(st.tcount && st.tstep && st.n_episodes && st.win_count && st.len_sum && st.score_sum, "crl_blokus_rollout: NULL stats pointer");
    CRL_REQUIRE(T >= 0 && T <= (1 << 24), "crl_blokus_rollout: T=%d out of range", T);
    if (T == 0) return CRL_OK;
    hipLaunchKernelGGL(blokus_rollout_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const BlkTables *)ctx->blokus, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T,
                       occ, inv, score, round, to_move, st);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

} // extern "C"
