/*
 * zamd_inflate_impl.h — batch inflate kernels and their host entry points.
 * Included by zamd_engine.hip (one translation unit, shares zamd_engine).
 *
 * Work unit: a SEGMENT of a raw deflate stream — a bit position where a block
 * starts, the byte where the readable input ends, and where its output goes.
 * One wavefront decodes one segment (k_inf_decode):
 *   - the compressed bytes are held 2 x 256 B at a time in two VGPRs (one dword
 *     per lane) and picked out with v_readlane, so the bit reader makes no
 *     memory access per symbol (the speculative decode of 64 bit offsets at once
 *     reads a copy of the same 512 bytes in LDS, three dwords per lane);
 *   - Huffman tables are per-wave LDS images: a 2^9 (literal/length) or 2^8
 *     (distance) first-level table of u32 {code length, extra bits, kind, flags,
 *     value}; longer codes fall back to the canonical first-code search;
 *   - every lane decodes the token that would start at its own bit offset of a
 *     64-bit span and notes where the token behind it would start; the wave walks
 *     the real tokens through those notes (one v_readlane a token) and parks them
 *     one per lane; every 64 tokens it scans their lengths, stores the literals
 *     and copies the matches through a stage in LDS.
 * Output is first written as 16-bit elements: values >= 256 name a byte of the
 * 32 KiB of output in front of the segment, which another wave is producing at
 * the same time.  k_inf_tail_* then resolve the last 32 KiB of every segment in
 * stream order and k_inf_final resolves everything else in parallel and writes
 * the bytes.
 *
 * Semantics follow inflate.c:605-1252 (block headers :836-895, dynamic tables
 * :896-1003, codes :1005-1200) and inftrees.c:32-306 (validity rules of code
 * length sets); error numbers map to the reference's message strings.
 */

/* Table and stage sizes set the LDS per wavefront (6.6 KB as they are: room for the 20 wavefronts a CU runs of this
 * kernel, not for many more).  Measured when the kernel ran 27 wavefronts to a CU: 10 root bits and a stage of 1 024
 * elements (9 KB, 17 wavefronts) decoded 8 GiB at 52 GB/s, 9 bits and 512 elements at 56; 8 root bits send too many
 * symbols to the search for longer codes (41 GB/s). */
#ifndef ZI_ROOT_L
#define ZI_ROOT_L 9
#endif
#ifndef ZI_ROOT_D
#define ZI_ROOT_D 8
#endif
#define ZI_ROOT_C 7
#ifndef ZI_SPAN_REGS
#define ZI_SPAN_LDS 1        /* zi_span's lanes pick their input dwords from a copy of the window in LDS: three reads instead of ten v_readlane and
                              * the selects around them (text 2 GiB: decode 45.2 -> 40.6 ms, mixed 8 GiB 136.2 -> 134.3; -DZI_SPAN_REGS builds the other form) */
#endif
#define ZI_MARK   256u       /* first marker value: byte 0 of the preceding 32 KiB */
#ifndef ZI_STAGE
#define ZI_STAGE  512u       /* elements of the per-wave output stage (>= 258) */
#endif
#ifndef ZI_SPLIT_SEGMENTS
#define ZI_SPLIT_SEGMENTS 20480u   /* zamd_inflate_split_dev cuts chunks into parts while that many segments are not exceeded: four times
                                    * the 5 120 wavefronts of k_inf_decode that the GPU holds */
#define ZI_SPLIT_CHUNKS   5120u    /* ... and while the chunks alone do not fill those */
#endif
#define ZI_MARKS  15u        /* block boundaries a segment can report (zamd_inflate_block_marks); a row of the result has one more entry */
#define ZI_EMPTY  0xffffu    /* stage element not produced yet (no byte or marker has this value) */
__constant__ uint8_t zi_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   /* inflate.c:608-609 */

enum {
    ZI_OK_END = 0, ZI_OK_FINAL = 1, ZI_OK_SYNC = 2, ZI_NEED_INPUT = 3,
    ZI_ERR_BTYPE = 16, ZI_ERR_STORED, ZI_ERR_TOOMANY, ZI_ERR_CLSET, ZI_ERR_REPEAT, ZI_ERR_NOEOB,
    ZI_ERR_LLSET, ZI_ERR_DSET, ZI_ERR_LCODE, ZI_ERR_DCODE, ZI_ERR_FAR, ZI_ERR_OUTCAP, ZI_ERR_LENGTH
};
#define ZI_MODE_SYNCSTOP 1u
#define ZI_MODE_EXACT    2u
#define ZI_MODE_BLOCKSTOP 4u

static const char *zi_message(uint32_t status)
{
    switch (status) {   /* strings of inflate.c:859, 868, 907, 929, 952/975, 989, 1003, 1012, inffast.c:273/288, 180/190 */
    case ZI_ERR_BTYPE:   return "invalid block type";
    case ZI_ERR_STORED:  return "invalid stored block lengths";
    case ZI_ERR_TOOMANY: return "too many length or distance symbols";
    case ZI_ERR_CLSET:   return "invalid code lengths set";
    case ZI_ERR_REPEAT:  return "invalid bit length repeat";
    case ZI_ERR_NOEOB:   return "invalid code -- missing end-of-block";
    case ZI_ERR_LLSET:   return "invalid literal/lengths set";
    case ZI_ERR_DSET:    return "invalid distances set";
    case ZI_ERR_LCODE:   return "invalid literal/length code";
    case ZI_ERR_DCODE:   return "invalid distance code";
    case ZI_ERR_FAR:     return "invalid distance too far back";
    case ZI_ERR_OUTCAP:  return "output exceeds the space given for the segment";
    case ZI_ERR_LENGTH:  return "segment did not inflate to the indexed length";
    case ZI_NEED_INPUT:  return "compressed data ends inside a block";
    default:             return "";
    }
}

/* layout shared with zamd_inflate_seg in zamd.h */
struct ZiSeg {
    uint64_t in_bit, in_end_bit, out_off, out_cap;
    uint32_t win_avail, mode;
    uint64_t out_len, end_bit;
    uint32_t status, final_seen;
    uint64_t stop_bit;
};

/* Table entry: bits 0-3 code length (0 = no code this short), 4-7 number of extra
 * bits, 8-9 kind, 16-31 value.  Literal/length code: kind 0 literal (value =
 * byte), 1 length (value = base length), 2 end of block, 3 invalid symbol.
 * Distance code: kind 0 (value = base distance) or 3.  Code-length code: value =
 * symbol.  Bits 10 and 11: ZI_E_ODD, ZI_E_EOB. */
#define ZI_K_LIT 0u
#define ZI_K_LEN 1u
#define ZI_K_EOB 2u
#define ZI_K_BAD 3u
enum { ZI_T_C = 0, ZI_T_L = 1, ZI_T_D = 2 };
#define ZI_E_ODD 0x400u      /* entry of a first-level table that the speculative decode leaves to the one-token path: no code this short, or an invalid symbol */
#define ZI_E_EOB 0x800u      /* the end-of-block code */

__device__ __forceinline__ uint32_t zi_entry(uint32_t table, uint32_t sym, uint32_t len)
{
    if (table == ZI_T_L) {
        if (sym < 256) return len | (ZI_K_LIT << 8) | (sym << 16);
        if (sym == 256) return len | (ZI_K_EOB << 8) | ZI_E_EOB;
        if (sym > 285) return len | (ZI_K_BAD << 8) | ZI_E_ODD;
        const uint32_t lc = sym - 257u;
        return len | (zc_len_extra(lc) << 4) | (ZI_K_LEN << 8) | ((zc_len_base(lc) + 3u) << 16);
    }
    if (table == ZI_T_D) {
        if (sym >= 30) return len | (ZI_K_BAD << 8) | ZI_E_ODD;
        return len | (zc_dist_extra(sym) << 4) | ((zc_dist_base(sym) + 1u) << 16);
    }
    return len | (sym << 16);
}

struct ZiTab {                 /* one canonical code: first-level table + search arrays */
    uint32_t *lut; uint16_t *sorted;
    uint16_t *first, *cnt, *off;   /* indexed by code length 1..15 */
    uint32_t table;
};

struct ZiLds {
    uint32_t lut_l[1u << ZI_ROOT_L], lut_d[1u << ZI_ROOT_D], lut_c[1u << ZI_ROOT_C];
    uint16_t sorted_l[288], sorted_d[32], sorted_c[20];
    uint16_t meta[3][3][16];
    uint32_t cntw[16];
    uint8_t lens[320 + 8];
    uint8_t cl[20 + 4];
    __attribute__((aligned(16))) uint16_t stage[ZI_STAGE];      /* output of the tokens being written, before it goes to memory */
#ifdef ZI_SPAN_LDS
    uint32_t ring[128 + 2];        /* the 512 input bytes the two window registers hold, where zi_span's lanes pick their own dwords
                                    * (the first two once more behind the end: three dwords in a row never wrap) */
#endif
    /* block marks of a counting pass (zi_mark): the records, how many, the spacing and the output position the next one waits for */
    uint64_t bm[ZI_MARKS + 1];
    uint64_t bm_next;
    uint32_t bm_n, bm_step;
};

struct ZiBits {
#ifdef ZI_SPAN_LDS
    uint32_t *ring;
#endif
    const uint32_t *base; uint64_t ndw;   /* readable dwords */
    uint32_t cur, nxt;                    /* per lane: dword wbase+lane, wbase+64+lane */
    uint64_t wbase;
    uint64_t bitpos;
};

__device__ __forceinline__ uint32_t zi_ldw(const ZiBits &b, uint64_t i) { return i < b.ndw ? b.base[i] : 0u; }

#ifdef ZI_SPAN_LDS
struct ZiBits;
__device__ __forceinline__ void zi_ring_put(ZiBits &b, uint32_t lane);
#endif

__device__ __forceinline__ void zi_bits_init(ZiBits &b, const uint32_t *base, uint64_t ndw, uint64_t bitpos, uint32_t lane)
{
    b.base = base; b.ndw = ndw; b.bitpos = bitpos;
    b.wbase = (bitpos >> 5) & ~63ull;
    b.cur = zi_ldw(b, b.wbase + lane);
    b.nxt = zi_ldw(b, b.wbase + 64 + lane);
#ifdef ZI_SPAN_LDS
    zi_ring_put(b, lane);
#endif
}

#ifdef ZI_SPAN_LDS
__device__ __forceinline__ void zi_ring_put(ZiBits &b, uint32_t lane)
{
    const uint32_t i0 = ((uint32_t)b.wbase + lane) & 127u, i1 = i0 ^ 64u;       /* wbase is a multiple of 64 */
    b.ring[i0] = b.cur; b.ring[i1] = b.nxt;
    if (lane < 2u) b.ring[128u + lane] = i0 < 2u ? b.cur : b.nxt;
}
#endif

/* >= 33 valid bits at the current position (wave-uniform result) */
__device__ __forceinline__ uint64_t zi_peek(ZiBits &b, uint32_t lane)
{
    uint64_t dw = (b.bitpos >> 5) - b.wbase;
    while (dw >= 64) {
        b.cur = b.nxt; b.wbase += 64; b.nxt = zi_ldw(b, b.wbase + 64 + lane); dw -= 64;
        if (dw >= 64) {       /* a jump over more than one window (stored blocks) */
            b.wbase = (b.bitpos >> 5) & ~63ull;
            b.cur = zi_ldw(b, b.wbase + lane); b.nxt = zi_ldw(b, b.wbase + 64 + lane);
            dw = (b.bitpos >> 5) - b.wbase;
        }
#ifdef ZI_SPAN_LDS
        zi_ring_put(b, lane);
#endif
    }
    uint32_t d0 = (uint32_t)dw;
    uint32_t lo = __builtin_amdgcn_readlane(b.cur, d0);
    uint32_t hi1 = __builtin_amdgcn_readlane(b.cur, (d0 + 1) & 63u);
    uint32_t hi2 = __builtin_amdgcn_readlane(b.nxt, 0);
    uint32_t hi = d0 == 63 ? hi2 : hi1;
    return (((uint64_t)hi << 32) | lo) >> (b.bitpos & 31u);
}

__device__ __forceinline__ uint32_t zi_uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

/* v with lane `sel` replaced by the wave-uniform value x.  Two vector
 * instructions on purpose: the scalar unit, shared by the four SIMDs of a CU, is
 * the scarce resource of the decoder. */
__device__ __forceinline__ uint32_t zi_writelane(uint32_t v, uint32_t x, uint32_t sel)
{
    return (threadIdx.x == sel) ? x : v;
}

/* Inclusive prefix sum over the 64 lanes with DPP row shifts and broadcasts (no LDS). */
__device__ __forceinline__ uint32_t zi_scan_incl(uint32_t v)
{
    uint32_t s = v;
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);    /* row_shr:3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, true);    /* row_shr:4, banks 1-3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, true);    /* row_shr:8, banks 2-3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, true);    /* row_bcast:15 into rows 1, 3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, true);    /* row_bcast:31 into rows 2, 3 */
    return s;
}

/* Lane l decodes the token that would start l bits behind bit position sb:
 * tlen = its length in bits (bits 0-7) with a note in bits 8+ (1 = end of block,
 * 2 = leave it to the one-token-at-a-time path: long code or invalid symbol),
 * word = the parked-symbol word.  Only lanes on the true token chain matter. */
#define ZI_W_PAIR    0x40000000u     /* parked-symbol word of two literals: first byte in bits 0-7, second in bits 8-15 */
#define ZI_SP_EOB    1u
#define ZI_SP_SCALAR 2u
__device__ __forceinline__ void zi_span(ZiBits &b, uint64_t sb, uint32_t lane, const uint32_t *lut_l, const uint32_t *lut_d,
                                        uint32_t &tlen, uint32_t &word, bool pairs)
{
    b.bitpos = sb;
    (void)zi_peek(b, lane);                              /* brings the window up to sb */
#ifdef ZI_SPAN_LDS
    const uint32_t t = ((uint32_t)sb & 31u) + lane, kk = t >> 5, sh = t & 31u;
    const uint32_t dwi = (uint32_t)(sb >> 5) + kk;        /* the window holds dwords wbase .. wbase + 127, and sb lies in its first half */
    const uint32_t *const rp = b.ring + (dwi & 127u);
    const uint32_t A = rp[0], Bv = rp[1], C = rp[2];
#else
    const uint32_t d0 = (uint32_t)((sb >> 5) - b.wbase);  /* 0..63 */
    uint32_t D[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const uint32_t a = __builtin_amdgcn_readlane(b.cur, (d0 + i) & 63u), c = __builtin_amdgcn_readlane(b.nxt, (d0 + i) & 63u);
        D[i] = d0 + i < 64 ? a : c;
    }
    const uint32_t t = ((uint32_t)sb & 31u) + lane, kk = t >> 5, sh = t & 31u;
    const uint32_t A = kk == 0 ? D[0] : kk == 1 ? D[1] : D[2];
    const uint32_t Bv = kk == 0 ? D[1] : kk == 1 ? D[2] : D[3];
    const uint32_t C = kk == 0 ? D[2] : kk == 1 ? D[3] : D[4];
#endif
    const uint32_t lo = __builtin_amdgcn_alignbit(Bv, A, sh), hi = __builtin_amdgcn_alignbit(C, Bv, sh);
    const uint64_t W = ((uint64_t)hi << 32) | lo;
    const uint32_t e1 = lut_l[lo & ((1u << ZI_ROOT_L) - 1u)];
    const uint32_t l1 = e1 & 15u, lx = (e1 >> 4) & 15u, k1 = (e1 >> 8) & 3u, v1 = e1 >> 16;
    const uint32_t o2 = l1 + lx;
    const uint32_t mlen = v1 + ((lo >> l1) & ((1u << lx) - 1u));
    const uint32_t e2 = lut_d[(lo >> o2) & ((1u << ZI_ROOT_D) - 1u)];
    const uint32_t l2 = e2 & 15u, dx = (e2 >> 4) & 15u, v2 = e2 >> 16;
    const uint32_t o3 = o2 + l2;
    const uint32_t dist = v2 + ((uint32_t)(W >> o3) & ((1u << dx) - 1u));
    const bool islen = k1 == ZI_K_LEN;
    const bool scalar = ((e1 | (islen ? e2 : 0u)) & ZI_E_ODD) != 0u;
    tlen = islen ? o3 + dx : l1;
    word = islen ? (0x80000000u | ((mlen - 3u) << 16) | dist) : v1;
    if (pairs) {
        /* a literal followed by a literal is one token of two bytes: the chain of tokens, which is followed one at a
         * time, gets shorter where literals come in rows (wave-uniform choice: the second look-up costs every lane) */
        const uint32_t eb = lut_l[(lo >> l1) & ((1u << ZI_ROOT_L) - 1u)];
        const uint32_t lb = eb & 15u;
        if (((e1 | eb) & 0xf00u) == 0u) {                /* both are literals: kind 0, neither flag */ tlen = l1 + lb; word = ZI_W_PAIR | ((eb >> 16) << 8) | v1; }
    }
    tlen |= (e1 & ZI_E_EOB) >> 3;                        /* ZI_SP_EOB << 8 */
    if (scalar) tlen = ZI_SP_SCALAR << 8;
}

/* Builds the tables of one code from lens[0..n).  Returns 0 = complete, 1 =
 * over-subscribed, 2 = incomplete; *maxlen = longest code (0 = no codes).
 * (Validity rules: inftrees.c:107-138.) */
__device__ uint32_t zi_build(ZiLds &L, const uint8_t *lens, uint32_t n, uint32_t root, ZiTab T, uint32_t lane, uint32_t *maxlen)
{
    if (lane < 16) L.cntw[lane] = 0;
    __syncthreads();
    for (uint32_t s = lane; s < n; s += 64) atomicAdd(&L.cntw[lens[s]], 1u);
    __syncthreads();
    uint32_t cntv = lane < 16 ? L.cntw[lane] : 0u;      /* lane l: number of codes of length l */
    int32_t left = 1; uint32_t code = 0, offs = 0, mx = 0, over = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        uint32_t c = __builtin_amdgcn_readlane(cntv, l);
        left = (left << 1) - (int32_t)c;
        if (left < 0) { over = 1; left = 0; }
        if (c) mx = l;
        if (lane == 0) { T.first[l] = (uint16_t)code; T.cnt[l] = (uint16_t)c; T.off[l] = (uint16_t)offs; }
        code = (code + c) << 1;
        offs += c;
    }
    *maxlen = mx;
    __syncthreads();
    if (over) return 1;
    /* stable placement by (length, symbol): 64 symbols per round */
    uint32_t runv = 0;                                   /* lane l: codes of length l placed so far */
    for (uint32_t s0 = 0; s0 < n; s0 += 64) {
        uint32_t s = s0 + lane;
        uint32_t myl = s < n ? lens[s] : 0u;
        uint32_t pos = 0;
        for (uint32_t l = 1; l <= mx; l++) {
            uint64_t bal = __ballot(myl == l);
            if (bal == 0) continue;
            uint32_t base = __builtin_amdgcn_readlane(runv, l);
            if (myl == l) pos = base + __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == l) runv += __popcll(bal);
        }
        if (myl) T.sorted[T.off[myl] + pos] = (uint16_t)s;
    }
    __syncthreads();
    for (uint32_t i = lane; i < (1u << root); i += 64) {
        uint32_t rev = __brev(i) >> (32 - root);
        uint32_t e = 0;
        uint32_t lim = mx < root ? mx : root;
        for (uint32_t l = 1; l <= lim; l++) {
            uint32_t d = (rev >> (root - l)) - T.first[l];
            if (d < T.cnt[l]) { e = zi_entry(T.table, T.sorted[T.off[l] + d], l); break; }
        }
        T.lut[i] = e ? e : (uint32_t)ZI_E_ODD;
    }
    __syncthreads();
    return left > 0 ? 2u : 0u;
}

/* Code longer than the first-level table: canonical search.  Returns the table
 * entry, or 0 when no code matches. */
__device__ __forceinline__ uint32_t zi_slow(const ZiTab &T, uint32_t bits, uint32_t root)
{
    uint32_t rev = __brev(bits) >> 17;                   /* first 15 stream bits, first bit on top */
    for (uint32_t l = root + 1; l <= 15; l++) {
        uint32_t d = (rev >> (15 - l)) - zi_uni(T.first[l]);
        if (d < zi_uni(T.cnt[l])) return zi_entry(T.table, zi_uni(T.sorted[zi_uni(T.off[l]) + d]), l);
    }
    return 0u;
}

/* Notes the block boundary at `rel_bit` bits behind the segment's start, where `pos` bytes have been produced: one record
 * every bm_step bytes of output at least.  A full list keeps every other record and doubles the spacing, so a segment
 * of any length reports boundaries spread over all of it.  All values are the same in every lane. */
__device__ __noinline__ void zi_mark(ZiLds &L, uint64_t rel_bit, uint64_t pos, uint32_t lane)
{
    if ((rel_bit | pos) >> 32) return;
    uint32_t n = L.bm_n, step = L.bm_step;
    if (n == ZI_MARKS) {
        const uint64_t v = lane < ZI_MARKS / 2u ? L.bm[2u * lane + 1u] : 0ull;
        __syncthreads();
        if (lane < ZI_MARKS / 2u) L.bm[lane] = v;
        n = ZI_MARKS / 2u;
        step = step < 0x40000000u ? step * 2u : step;
    }
    __syncthreads();
    if (lane == 0) { L.bm[n] = (rel_bit << 32) | pos; L.bm_n = n + 1u; L.bm_step = step; L.bm_next = pos + step; }
    __syncthreads();
}

template <bool WRITE>
/* five wavefronts to a SIMD: the kernel takes 92 registers and spills nothing.  Six (80 registers, two spilled values; the
 * LDS has room for no more) were 3 % faster while the walk along the tokens took twelve instructions a token, and are 6 %
 * slower now (8 GiB mixed: 122.4 against 115.0 ms); four are as fast as five on small calls and slower on large ones */
#ifndef ZI_WAVES_PER_SIMD
#define ZI_WAVES_PER_SIMD 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ZI_WAVES_PER_SIMD, ZI_WAVES_PER_SIMD))) void k_inf_decode(const uint8_t *comp, uint64_t comp_bytes, ZiSeg *segs, uint32_t nseg, uint16_t *out16,
                                                                                                                                 uint64_t *bmarks, uint32_t mark_step, uint32_t pair_mode, const uint32_t *order)
{
    __shared__ ZiLds L;
    const uint32_t lane = threadIdx.x;
    const uint32_t si = order ? order[blockIdx.x] : blockIdx.x;     /* (k_inf_order: the long segments start first) */
    if (si >= nseg) return;
    ZiSeg S = segs[si];
    const bool marking = !WRITE && bmarks != nullptr;        /* only a counting pass reports block marks */
    if (marking) {
        if (lane == 0) { L.bm_n = 0; L.bm_step = mark_step; L.bm_next = mark_step; }
        __syncthreads();
    }
    const uint64_t end_bits = S.in_end_bit;
    ZiBits B;
#ifdef ZI_SPAN_LDS
    B.ring = L.ring;
#endif
    zi_bits_init(B, (const uint32_t *)comp, (comp_bytes + 3) >> 2, S.in_bit, lane);
    ZiTab TL{L.lut_l, L.sorted_l, L.meta[0][0], L.meta[0][1], L.meta[0][2], ZI_T_L};
    ZiTab TD{L.lut_d, L.sorted_d, L.meta[1][0], L.meta[1][1], L.meta[1][2], ZI_T_D};
    ZiTab TC{L.lut_c, L.sorted_c, L.meta[2][0], L.meta[2][1], L.meta[2][2], ZI_T_C};
    uint16_t *dst = out16 + S.out_off;

    uint64_t pos = 0;                  /* elements produced so far */
    uint64_t good_bit = S.in_bit, good_pos = 0;
    uint32_t status = 0xffffffffu, final_seen = 0;
    bool fixed_ready = false;
    uint32_t mysym = 0, k = 0;
    bool pairs = pair_mode == 1u;
    uint32_t flushed_bit = (uint32_t)S.in_bit;       /* where the last flush stood in the input (low bits) */

    /* flushes the parked symbols; returns false on error (status set) */
    auto flush = [&]() -> bool {
        if (k == 0) return true;
        if (B.bitpos > end_bits) { status = ZI_NEED_INPUT; return false; }   /* some parked symbols came from beyond the input */
        const bool valid = lane < k;
        const uint32_t w = mysym;
        const bool ism = valid && (w >> 31);
        const bool isp = valid && (w >> 30) == 1u;
        const uint32_t len = valid ? (ism ? ((w >> 16) & 0xffu) + 3u : isp ? 2u : 1u) : 0u;
        if (pair_mode == 2u) {
            /* literal pairs pay where tokens are short (many to a span of 64 bits: following them one at a time is then most
             * of the work) and cost where they are long (the second look-up of every lane): on below 8 bits a token on
             * average, off again from 14 (a pair counts as one token of twice the bits) */
            const uint32_t db = (uint32_t)B.bitpos - flushed_bit;
            flushed_bit = (uint32_t)B.bitpos;
            if (k >= 32u) pairs = db < k * (pairs ? 18u : 10u);
        }
        const uint32_t incl = zi_scan_incl(len);
        const uint32_t off = incl - len;
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        /* distance checks (inflate.c:1145-1158) */
        bool far = ism && ((int64_t)(pos + off) - (int64_t)(w & 0xffffu) < -(int64_t)S.win_avail);
        if (__ballot(far)) { status = ZI_ERR_FAR; return false; }
        if (pos + total > S.out_cap) { status = ZI_ERR_OUTCAP; return false; }
        if (WRITE) {
            /* Every lane writes its own token.  The tokens of a sub-batch (at most
             * ZI_STAGE elements) are assembled in LDS: a copy whose source lies inside
             * the sub-batch waits, element by element, until the owner has produced it
             * (sources lie strictly below their destinations, so the lowest unfinished
             * element can always advance); sources below the sub-batch come from
             * memory, four elements at a time, or are window markers. */
            uint32_t done_tok = 0, base_off = 0;
            while (done_tok < k) {
                const bool in = valid && lane >= done_tok && (incl - base_off) <= ZI_STAGE;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(in));
                const uint32_t sub_total = __builtin_amdgcn_readlane(incl, done_tok + cnt - 1) - base_off;
                *(uint4 *)(L.stage + 8u * lane) = make_uint4(~0u, ~0u, ~0u, ~0u);      /* all of it ZI_EMPTY: one store */
                const uint32_t o = off - base_off;
                const int64_t bs = (int64_t)(pos + base_off);          /* position of stage[0] in the segment */
                const int32_t sr = (int32_t)o - (int32_t)(w & 0xffffu);
                uint32_t rem = (in && ism) ? len : 0u, j = 0;
                if (in && !ism) { L.stage[o] = (uint16_t)(w & 0xffu); if (isp) L.stage[o + 1u] = (uint16_t)((w >> 8) & 0xffu); }
                /* long or self-overlapping copies are done by all lanes together, each when
                 * it is the lowest unfinished token (everything below it is complete then) */
                const bool mycoop = rem != 0 && ((w & 0xffffu) < len || len > 32u);
                const uint64_t coopm = __ballot(mycoop);
                for (;;) {
                    const uint64_t um = __ballot(rem != 0);
                    if (um == 0) break;
                    const uint32_t first = (uint32_t)__builtin_ctzll(um);
                    if ((coopm >> first) & 1ull) {
                        const uint32_t wc = __builtin_amdgcn_readlane(w, first), oc = __builtin_amdgcn_readlane(o, first);
                        const uint32_t lc = ((wc >> 16) & 0xffu) + 3u, dc = wc & 0xffffu;
                        const int32_t src = (int32_t)oc - (int32_t)dc;
                        for (uint32_t jj = lane; jj < lc; jj += 64) {
                            const int32_t sidx = src + (int32_t)(dc >= lc ? jj : jj % dc);
                            uint32_t v;
                            if (sidx < 0) { const int64_t a = bs + sidx; v = a < 0 ? (uint32_t)(ZI_MARK + 32768 + a) : (uint32_t)dst[a]; }
                            else v = L.stage[sidx];
                            L.stage[oc + jj] = (uint16_t)v;
                        }
                        if (lane == first) rem = 0;
                        continue;
                    }
                    if (rem && !mycoop) {
                        const int32_t sidx = sr + (int32_t)j;
                        if (sidx + 7 < 0) {
                            /* eight elements whose sources all lie below the sub-batch: one 16-byte load (a trip to memory
                             * is what an iteration of this loop costs) */
                            const int64_t a = bs + sidx;
                            uint32_t v[8];
                            if (a >= 0) {
                                uint4 x;
                                __builtin_memcpy(&x, dst + a, 16);
                                v[0] = x.x & 0xffffu; v[1] = x.x >> 16; v[2] = x.y & 0xffffu; v[3] = x.y >> 16;
                                v[4] = x.z & 0xffffu; v[5] = x.z >> 16; v[6] = x.w & 0xffffu; v[7] = x.w >> 16;
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; i++) v[i] = (a + i < 0) ? (uint32_t)(ZI_MARK + 32768 + (a + i)) : (uint32_t)dst[a + i];
                            }
                            const uint32_t take = rem < 8u ? rem : 8u;
#pragma unroll
                            for (uint32_t i = 0; i < 8; i++) if (i < take) L.stage[o + j + i] = (uint16_t)v[i];
                            j += take; rem -= take;
                        } else {
                            uint32_t v;
                            if (sidx < 0) { const int64_t a = bs + sidx; v = a < 0 ? (uint32_t)(ZI_MARK + 32768 + a) : (uint32_t)dst[a]; }
                            else v = L.stage[sidx];
                            if (v != ZI_EMPTY) { L.stage[o + j] = (uint16_t)v; j++; rem--; }
                        }
                    }
                }
                {
                    /* eight elements a lane: one 16-byte store where all eight belong to the sub-batch (the addresses are
                     * 2-byte aligned only), single elements at its end */
                    const uint32_t j0 = 8u * lane;
                    if (j0 + 8u <= sub_total) {
                        const uint4 v = *(const uint4 *)(L.stage + j0);
                        __builtin_memcpy(dst + bs + j0, &v, 16);
                    } else {
                        for (uint32_t jj = j0; jj < sub_total; jj++) dst[bs + jj] = L.stage[jj];
                    }
                }
                __threadfence_block();                                 /* later sub-batches and batches read these */
                done_tok += cnt; base_off += sub_total;
            }
        }
        pos += total; k = 0;
        return true;
    };
    /* an error only counts when every bit that led to it lies inside the input */
    auto fail = [&](uint32_t code, uint32_t ahead) { status = (B.bitpos + ahead > end_bits) ? (uint32_t)ZI_NEED_INPUT : code; };

    while (status == 0xffffffffu) {
        if (B.bitpos + 3 > end_bits) { status = (B.bitpos == end_bits) ? (uint32_t)ZI_OK_END : (uint32_t)ZI_NEED_INPUT; break; }
        uint64_t bits = zi_peek(B, lane);
        const uint32_t last = (uint32_t)bits & 1u, type = ((uint32_t)bits >> 1) & 3u;
        B.bitpos += 3;
        bool empty_stored = false;
        if (type == 0) {
            B.bitpos = (B.bitpos + 7) & ~7ull;
            if (B.bitpos + 32 > end_bits) { status = ZI_NEED_INPUT; break; }
            bits = zi_peek(B, lane);
            const uint32_t sl = (uint32_t)bits & 0xffffu, nsl = ((uint32_t)bits >> 16) & 0xffffu;
            if (sl != (nsl ^ 0xffffu)) { fail(ZI_ERR_STORED, 32); break; }
            B.bitpos += 32;
            const uint64_t byte0 = B.bitpos >> 3;
            if ((byte0 + sl) * 8ull > end_bits) { status = ZI_NEED_INPUT; break; }
            if (pos + sl > S.out_cap) { status = ZI_ERR_OUTCAP; break; }
            if (WRITE) { for (uint32_t j = lane; j < sl; j += 64) dst[pos + j] = comp[byte0 + j]; __threadfence_block(); }
            pos += sl;
            B.bitpos += 8ull * sl;
            empty_stored = sl == 0;
        } else if (type == 3) {
            fail(ZI_ERR_BTYPE, 0); break;
        } else {
            if (type == 1) {
                if (!fixed_ready) {
                    for (uint32_t s = lane; s < 288; s += 64) L.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                    if (lane < 32) L.lens[288 + lane] = 5;
                    __syncthreads();
                    uint32_t mx;
                    zi_build(L, L.lens, 288, ZI_ROOT_L, TL, lane, &mx);
                    zi_build(L, L.lens + 288, 32, ZI_ROOT_D, TD, lane, &mx);
                    fixed_ready = true;
                }
            } else {
                fixed_ready = false;
                if (B.bitpos + 14 > end_bits) { status = ZI_NEED_INPUT; break; }
                const uint32_t nlen = ((uint32_t)bits >> 3 & 31u) + 257u, ndist = ((uint32_t)bits >> 8 & 31u) + 1u,
                               ncode = ((uint32_t)bits >> 13 & 15u) + 4u;
                B.bitpos += 14;
                if (nlen > 286 || ndist > 30) { fail(ZI_ERR_TOOMANY, 0); break; }
                if (lane < 19) L.cl[lane] = 0;
                __syncthreads();
                for (uint32_t i = 0; i < ncode;) {       /* 3-bit lengths, 11 per peek */
                    bits = zi_peek(B, lane);
                    uint32_t take = ncode - i < 11 ? ncode - i : 11;
                    if (lane < take) L.cl[zi_order[i + lane]] = (uint8_t)((bits >> (3 * lane)) & 7u);
                    B.bitpos += 3 * take; i += take;
                }
                __syncthreads();
                if (B.bitpos > end_bits) { status = ZI_NEED_INPUT; break; }
                uint32_t mx;
                uint32_t r = zi_build(L, L.cl, 19, ZI_ROOT_C, TC, lane, &mx);
                if (r != 0 && mx != 0) { fail(ZI_ERR_CLSET, 0); break; }      /* incomplete sets are invalid for this code */
                const uint32_t total = nlen + ndist;
                uint32_t idx = 0;
                bool bad = false;
                while (idx < total) {
                    bits = zi_peek(B, lane);
                    uint32_t e = zi_uni(L.lut_c[(uint32_t)bits & 127u]);
                    uint32_t l = e & 15u, sym = e >> 16;
                    if (l == 0) { l = 1; sym = 0; }                          /* empty code: inftrees.c:118-125 */
                    bits >>= l; B.bitpos += l;
                    if (sym < 16) {
                        if (lane == 0) L.lens[idx] = (uint8_t)sym;
                        idx++;
                    } else {
                        uint32_t rep, val = 0;
                        if (sym == 16) {
                            rep = 3 + ((uint32_t)bits & 3u); B.bitpos += 2;
                            if (idx == 0) { fail(ZI_ERR_REPEAT, 0); bad = true; break; }
                            __syncthreads();
                            val = zi_uni(L.lens[idx - 1]);
                        } else if (sym == 17) { rep = 3 + ((uint32_t)bits & 7u); B.bitpos += 3; }
                        else { rep = 11 + ((uint32_t)bits & 127u); B.bitpos += 7; }
                        if (idx + rep > total) { fail(ZI_ERR_REPEAT, 0); bad = true; break; }
                        for (uint32_t j = lane; j < rep; j += 64) L.lens[idx + j] = (uint8_t)val;
                        idx += rep;
                    }
                    if (B.bitpos > end_bits + 64) { status = ZI_NEED_INPUT; bad = true; break; }
                }
                if (bad) break;
                __syncthreads();
                if (B.bitpos > end_bits) { status = ZI_NEED_INPUT; break; }
                if (zi_uni(L.lens[256]) == 0) { fail(ZI_ERR_NOEOB, 0); break; }
                /* distance lengths follow the literal/length lengths; move them to a fixed place */
                uint8_t dl = lane < ndist ? L.lens[nlen + lane] : 0;
                __syncthreads();
                if (lane < 32) L.lens[288 + lane] = dl;
                __syncthreads();
                r = zi_build(L, L.lens, nlen, ZI_ROOT_L, TL, lane, &mx);
                if (r == 1 || (r == 2 && mx != 1)) { fail(ZI_ERR_LLSET, 0); break; }
                r = zi_build(L, L.lens + 288, ndist, ZI_ROOT_D, TD, lane, &mx);
                if (r == 1 || (r == 2 && mx > 1)) { fail(ZI_ERR_DSET, 0); break; }
            }
            /* ---- symbols of the block: 64 bit offsets are decoded at once, then the
             * chain of real tokens is followed through the lanes' results */
            bool bad = false, eob = false;
            uint64_t sb = B.bitpos;
            uint32_t sp = 0;                      /* offset of the next token inside the span */
            while (!eob && !bad) {
                uint32_t tl, wd;
                zi_span(B, sb, lane, L.lut_l, L.lut_d, tl, wd, pairs);
                const uint32_t to = tl >= 256u ? 256u + lane : lane + tl;
                for (;;) {
                    /* the hot part: nothing but ordinary tokens.  The scalar loop only follows the token
                     * lengths and notes where tokens start; their words then move to the parking lanes
                     * k, k+1, ... all at once (a forward lane permute: each starting lane sends its word
                     * to the slot of its rank among the starting lanes) */
                    uint32_t t = 0;
                    if (sp < 64) {
                        /* every lane knows where the token behind its own would start (`to`), so a step of the walk is one
                         * v_readlane and one comparison; a token that is not an ordinary one points beyond 255 and ends the
                         * walk standing on it */
                        unsigned long long M = 0;
                        uint32_t s2 = sp;
                        do {
                            M |= 1ull << s2;
                            s2 = __builtin_amdgcn_readlane(to, s2);
                        } while (s2 < 64u);
                        if (s2 >= 256u) { s2 -= 256u; M &= ~(1ull << s2); t = __builtin_amdgcn_readlane(tl, s2); }
                        const uint32_t c = (uint32_t)__popcll(M);
                        if (c) {
                            if (k + c > 64u) { B.bitpos = sb + sp; if (!flush()) { bad = true; break; } }   /* makes room: k = 0 */
                            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
                            const bool starts = (M >> lane) & 1ull;
                            /* lanes that start no token send to a slot nobody reads */
                            const uint32_t spare = k + c < 64u ? k + c : 0u;
                            const uint32_t to = starts ? k + rank : spare;
                            const uint32_t got = (uint32_t)__builtin_amdgcn_ds_permute((int)(to << 2), (int)wd);
                            if (lane - k < c) mysym = got;
                            k += c; sp = s2;
                        }
                    }
                    if (k == 64) { B.bitpos = sb + sp; if (!flush()) { bad = true; break; } continue; }
                    if (sp >= 64) break;
                    B.bitpos = sb + sp;
                    if ((t >> 8) == ZI_SP_EOB) { B.bitpos += t & 0xffu; eob = true; break; }
                    /* one token the long way: codes beyond the first-level tables, invalid symbols */
                    bits = zi_peek(B, lane);
                    uint32_t e = zi_uni(L.lut_l[(uint32_t)bits & ((1u << ZI_ROOT_L) - 1u)]);
                    if ((e & 15u) == 0) e = zi_slow(TL, (uint32_t)bits, ZI_ROOT_L);
                    if (e == 0) { fail(ZI_ERR_LCODE, 1); bad = true; break; }
                    uint32_t l = e & 15u, kd = (e >> 8) & 3u;
                    B.bitpos += l;
                    if (kd == ZI_K_BAD) { fail(ZI_ERR_LCODE, 0); bad = true; break; }
                    if (kd == ZI_K_EOB) { eob = true; break; }
                    uint32_t word = e >> 16;
                    if (kd == ZI_K_LEN) {
                        const uint32_t lx = (e >> 4) & 15u;
                        const uint32_t mlen = (e >> 16) + ((uint32_t)(bits >> l) & ((1u << lx) - 1u));
                        B.bitpos += lx;
                        bits = zi_peek(B, lane);
                        e = zi_uni(L.lut_d[(uint32_t)bits & ((1u << ZI_ROOT_D) - 1u)]);
                        if ((e & 15u) == 0) e = zi_slow(TD, (uint32_t)bits, ZI_ROOT_D);
                        if (e == 0) { fail(ZI_ERR_DCODE, 1); bad = true; break; }
                        l = e & 15u;
                        if (((e >> 8) & 3u) == ZI_K_BAD) { B.bitpos += l; fail(ZI_ERR_DCODE, 0); bad = true; break; }
                        const uint32_t dx = (e >> 4) & 15u;
                        const uint32_t dist = (e >> 16) + ((uint32_t)(bits >> l) & ((1u << dx) - 1u));
                        B.bitpos += l + dx;
                        word = 0x80000000u | ((mlen - 3u) << 16) | dist;
                    }
                    mysym = zi_writelane(mysym, word, k);
                    if (++k == 64) { if (!flush()) { bad = true; break; } }
                    /* the walk goes on behind the token, in this span if the token ends inside it */
                    sp = (uint32_t)(B.bitpos - sb);
                    if (sp >= 64u) break;
                }
                sb += 64; sp -= 64;
            }
            if (bad) break;
        }
        if (!flush()) break;
        if (B.bitpos > end_bits) { status = ZI_NEED_INPUT; break; }
        good_bit = B.bitpos; good_pos = pos;
        if (last) { final_seen = 1; status = ZI_OK_FINAL; break; }
        if (marking && pos >= L.bm_next) zi_mark(L, good_bit - S.in_bit, pos, lane);
        if (empty_stored && (S.mode & ZI_MODE_SYNCSTOP)) { status = ZI_OK_SYNC; break; }
        if ((S.mode & ZI_MODE_BLOCKSTOP) && B.bitpos >= S.stop_bit) { status = ZI_OK_SYNC; break; }
    }
    if ((S.mode & ZI_MODE_EXACT) && status < 3 && good_pos != S.out_cap) status = ZI_ERR_LENGTH;
    if (lane == 0) {
        segs[si].out_len = good_pos; segs[si].end_bit = good_bit; segs[si].status = status; segs[si].final_seen = final_seen;
    }
    if (marking) {
        __syncthreads();
        if (lane <= ZI_MARKS) bmarks[(uint64_t)si * (ZI_MARKS + 1u) + lane] = lane < L.bm_n ? L.bm[lane] : 0ull;
    }
}

/* Segment table of a chunk-indexed stream: chunk i = comp[off[i], off[i+1]) ->
 * out[i * chunk_size ...). */
__global__ void k_inf_setup(const uint64_t *chunk_off, uint64_t nchunks, uint32_t chunk_size, uint64_t n_out, uint32_t head_avail, ZiSeg *segs)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    ZiSeg s;
    s.in_bit = chunk_off[i] * 8ull; s.in_end_bit = chunk_off[i + 1] * 8ull;
    s.out_off = i * (uint64_t)chunk_size;
    uint64_t rest = n_out > s.out_off ? n_out - s.out_off : 0;
    s.out_cap = rest < chunk_size ? rest : chunk_size;
    uint64_t before = s.out_off + head_avail;
    s.win_avail = (uint32_t)(before < 32768 ? before : 32768);
    s.mode = ZI_MODE_EXACT;
    s.out_len = 0; s.end_bit = 0; s.status = 0xffffffffu; s.final_seen = 0; s.stop_bit = 0;
    segs[i] = s;
}

/* result[0] = index of the first segment whose status is not a success, or ~0 */
/* A launch of more segments than the GPU holds wavefronts ends with the segments that were started last: the order of
 * the workgroups is the order of the segments by their input length, longest first (in 1024 classes of length; the
 * order inside a class is whatever the atomics make it, which changes nothing in what is written where). */
__global__ __launch_bounds__(1024) void k_inf_order(const ZiSeg *segs, uint32_t nseg, uint32_t *order)
{
    __shared__ uint32_t cnt[1024];
    __shared__ unsigned long long longest;
    const uint32_t t = threadIdx.x;
    cnt[t] = 0;
    if (t == 0) longest = 0;
    __syncthreads();
    for (uint32_t i = t; i < nseg; i += 1024u) atomicMax(&longest, (unsigned long long)(segs[i].in_end_bit - segs[i].in_bit));
    __syncthreads();
    const uint64_t m = longest;
    const int top = m ? 64 - __clzll((long long)m) : 0;
    const uint32_t shift = top > 10 ? (uint32_t)(top - 10) : 0u;
    for (uint32_t i = t; i < nseg; i += 1024u) {
        const uint64_t c = (segs[i].in_end_bit - segs[i].in_bit) >> shift;
        atomicAdd(&cnt[1023u - (uint32_t)(c < 1023u ? c : 1023u)], 1u);
    }
    __syncthreads();
    if (t == 0) { uint32_t run = 0; for (uint32_t b = 0; b < 1024u; b++) { const uint32_t c = cnt[b]; cnt[b] = run; run += c; } }
    __syncthreads();
    for (uint32_t i = t; i < nseg; i += 1024u) {
        const uint64_t c = (segs[i].in_end_bit - segs[i].in_bit) >> shift;
        order[atomicAdd(&cnt[1023u - (uint32_t)(c < 1023u ? c : 1023u)], 1u)] = i;
    }
}

__global__ __launch_bounds__(1024) void k_inf_status(const ZiSeg *segs, uint64_t nseg, unsigned long long *result)
{
    unsigned long long best = ~0ull;
    for (uint64_t i = threadIdx.x; i < nseg; i += 1024) if (segs[i].status >= 3 && i < best) best = i;
    if (best != ~0ull) atomicMin(result, best);
}

/* Index into the 32 KiB in front of a segment that a window marker stands for.  The decoder writes no element above
 * ZI_MARK + 32767; the bound keeps every address the resolve step forms inside its buffers whatever the stage holds. */
__device__ __forceinline__ uint32_t zi_mark_index(uint32_t v) { const uint32_t i = v - ZI_MARK; return i < 32767u ? i : 32767u; }

__device__ __forceinline__ uint8_t zi_window_byte(const ZiSeg &S, uint32_t marker, const uint8_t *out8, const uint8_t *head)
{
    int64_t q = (int64_t)S.out_off - 32768 + (int64_t)zi_mark_index(marker);
    if (q >= 0) return __hip_atomic_load(out8 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return head[32768 + q];
}

/* The same for k_inf_final: the byte lies in a tail (or in the window in front of everything), which an earlier kernel has
 * written, so an ordinary load serves — sixteen of them in a row from one or two cache lines, where sixteen atomic loads
 * each went to the L2 by themselves (text, whose segments begin with 32 KiB full of references: 16.3 ms per 8 GiB). */
__device__ __forceinline__ uint8_t zi_window_byte_written(const ZiSeg &S, uint32_t marker, const uint8_t *out8, const uint8_t *head)
{
    const int64_t q = (int64_t)S.out_off - 32768 + (int64_t)zi_mark_index(marker);
    return q >= 0 ? out8[q] : head[32768 + q];
}

/* Last 32 KiB of every segment: plain bytes are written out, markers are
 * collected per segment as (offset in the tail) | (window index << 16). */
__global__ __launch_bounds__(1024) void k_inf_tail_pre(const ZiSeg *segs, const uint16_t *out16, uint8_t *out8, uint32_t *marks, uint32_t *mlist)
{
    const uint32_t si = blockIdx.x >> 5, part = blockIdx.x & 31u;
    const ZiSeg S = segs[si];
    const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
    const uint64_t e = (uint64_t)part * 1024 + threadIdx.x;
    uint32_t mk = 0, v = 0;
    if (e < tl) {
        const uint64_t p = S.out_off + S.out_len - tl + e;
        v = out16[p];
        if (v < ZI_MARK) out8[p] = (uint8_t)v; else mk = 1;
    }
    const uint64_t b = __ballot(mk);
    if (b) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&marks[si], (uint32_t)__popcll(b));
        base = __builtin_amdgcn_readfirstlane(base);
        if (mk) mlist[(uint64_t)si * 32768u + base + __popcll(b & ((1ull << lane) - 1ull))] = (uint32_t)e | (zi_mark_index(v) << 16);
    }
}

/* Markers inside those tails, in stream order: each needs the (already resolved)
 * tail in front of it.  One workgroup walks the segments; only the collected
 * markers are touched. */
__global__ __launch_bounds__(1024) void k_inf_tail_serial(const ZiSeg *segs, uint64_t nseg, uint8_t *out8,
                                                          const uint32_t *marks, const uint32_t *mlist, const uint8_t *head)
{
    for (uint64_t si = 0; si < nseg; si++) {
        const uint32_t cnt = marks[si];
        if (cnt == 0) continue;
        const ZiSeg S = segs[si];
        const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
        const uint64_t t0 = S.out_off + S.out_len - tl;
        for (uint32_t k = threadIdx.x; k < cnt; k += 1024) {
            const uint32_t m = mlist[si * 32768u + k];
            out8[t0 + (m & 0xffffu)] = zi_window_byte(S, ZI_MARK + (m >> 16), out8, head);
        }
        __threadfence_block();      /* one workgroup, one CU: its L1 sees its own stores */
        __syncthreads();
    }
}

/* The same by pointer jumping, for the regular case that every segment but the
 * last is at least 32 KiB long (its tail IS the next segment's window): a tail
 * element is a resolved byte (bit 31 set) or the slot of the window element it
 * copies, segment * 32768 + index; every round replaces a pointer by what it
 * points at, so chains that run back through s segments vanish in log2(s) rounds.
 * In-place updates are safe: a slot always holds either its byte or a pointer
 * to a slot that will hold the same byte. */
/* head == nullptr: the window in front of the first segment is not known yet (it is another rank's output): its
 * bytes stay references, bit 30 | index into that window, which the jump rounds treat as resolved. */
#define ZI_REF_BYTE 0x80000000u
#define ZI_REF_HEAD 0x40000000u
__global__ __launch_bounds__(1024) void k_inf_ref_init(const ZiSeg *segs, const uint16_t *out16, const uint8_t *head, uint32_t *ref)
{
    const uint32_t si = blockIdx.x >> 5, e = (blockIdx.x & 31u) * 1024u + threadIdx.x;
    const ZiSeg S = segs[si];
    const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
    uint32_t r = 0x80000000u;                            /* slots behind a short tail: every slot is read by the jump rounds */
    if (e < tl) {
        const uint32_t v = out16[S.out_off + S.out_len - tl + e];
        if (v < ZI_MARK) r = 0x80000000u | v;
        else {
            /* the byte lies less than 32 KiB in front of this segment, so inside the last 32 KiB of whichever segment holds it
             * (segments are back to back; short and empty ones are stepped over) or in the window in front of them all */
            const int64_t q = (int64_t)S.out_off - 32768 + (int64_t)zi_mark_index(v);
            int64_t sj = (int64_t)si - 1;
            while (sj >= 0 && (int64_t)segs[sj].out_off > q) sj--;
            if (sj < 0) r = head ? (ZI_REF_BYTE | head[32768 + q]) : (ZI_REF_HEAD | (uint32_t)(32768 + q));
            else {
                const uint64_t lj = segs[sj].out_len, tj = lj < 32768 ? lj : 32768;
                r = (uint32_t)sj * 32768u + (uint32_t)((uint64_t)q - (segs[sj].out_off + lj - tj));
            }
        }
    }
    ref[(uint64_t)si * 32768u + e] = r;
}
/* One jump round.  (Chains are as long as the data's history: in repetitive data nearly every group of 64 slots holds
 * a reference until the last rounds, so there is nothing to gain from skipping groups that hold none — measured.) */
/* open[round] is set by a round that leaves a reference behind; a round whose predecessor left none has nothing to do
 * (the rounds are launched for the longest chain the segment count allows; the mixed corpus needs 11 of the 14 at 8 192
 * segments: 11.0 -> 10.0 ms). */
/* A round follows a reference ZI_JUMP_HOPS steps, not one: every step reads a slot that holds its byte or a pointer
 * further along the same chain (whether or not another thread has advanced it meanwhile), so a chain shrinks to an
 * eighth per round at least and 8 192 segments need five rounds, not fourteen.  A round costs a pass over all slots
 * whatever it finds (0.23 ms per GiB of slots); the extra steps are taken by the slots that still hold references only. */
#define ZI_JUMP_HOPS 7u
#define ZI_JUMP_SHIFT 3u            /* log2(ZI_JUMP_HOPS + 1) */
__global__ __launch_bounds__(256) void k_inf_ref_jump(uint32_t *ref, uint64_t n, uint32_t *open, uint32_t round)
{
    if (round && !open[round - 1u]) return;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = ref[i];
    uint32_t nr = r;
    for (uint32_t h = 0; h < ZI_JUMP_HOPS && !(nr >> 30); h++) nr = __hip_atomic_load(ref + nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nr != r) ref[i] = nr;
    if (__ballot(!(nr >> 30)) && (threadIdx.x & 63u) == 0) open[round] = 1u;
}
/* The same for chunks that were cut into `use` parts, in three steps that touch most slots twice instead of once per
 * round: (1) part k = 1, 2, ... of every chunk takes over what its references into earlier parts of the SAME chunk hold,
 * which by then are bytes or references into earlier chunks (k_inf_ref_local, one launch per k); (2) the jump rounds run
 * over the slots of each chunk's last 32 KiB only, the only slots later chunks refer to (k_inf_ref_jump_last); (3)
 * every other slot takes over what its reference — into such a last 32 KiB — holds now (k_inf_ref_rest). */
__global__ __launch_bounds__(1024) void k_inf_ref_local(uint32_t *ref, uint32_t use, uint32_t k)
{
    const uint32_t chunk = blockIdx.x >> 5, si = chunk * use + k, e = (blockIdx.x & 31u) * 1024u + threadIdx.x;
    const uint64_t i = (uint64_t)si * 32768u + e;
    const uint32_t r = ref[i];
    if (r >> 30) return;
    if ((r >> 15) / use != chunk) return;                /* into an earlier chunk already */
    ref[i] = ref[r];                                     /* an earlier part: not written by this launch */
}
/* does slot e of segment S hold a byte of the last 32 KiB of its chunk (which ends at chunk_end) */
__device__ __forceinline__ bool zi_in_last_window(const ZiSeg &S, uint32_t e, uint64_t chunk_end)
{
    const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
    return e < tl && S.out_off + S.out_len - tl + e + 32768 >= chunk_end;
}
/* The groups of 1 024 slots that hold bytes of their chunk's last 32 KiB, listed once (a launch over all groups in every
 * round costs more in groups that have nothing to do than the rounds save) */
__global__ __launch_bounds__(256) void k_inf_ref_last_list(const ZiSeg *segs, uint64_t ngroups, uint32_t use, uint32_t chunk_size, uint64_t total,
                                                           uint2 *list, uint32_t cap, uint32_t *count)
{
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= ngroups) return;
    const uint32_t si = (uint32_t)(g >> 5), e0 = (uint32_t)(g & 31u) * 1024u;
    const ZiSeg S = segs[si];
    const uint64_t ce = (uint64_t)(si / use + 1u) * chunk_size, chunk_end = ce < total ? ce : total;
    const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
    if (e0 >= tl) return;
    const uint32_t e1 = (uint32_t)(e0 + 1024u < tl ? e0 + 1024u : tl);               /* slots [e0, e1) hold bytes */
    if (!zi_in_last_window(S, e1 - 1u, chunk_end)) return;
    /* the first of them that lies in the window (positions ascend with the slots) */
    const uint64_t p0 = S.out_off + S.out_len - tl + e0;
    const uint32_t lo = p0 + 32768 >= chunk_end ? 0u : (uint32_t)(chunk_end - 32768 - p0);
    const uint32_t at = atomicAdd(count, 1u);
    if (at < cap) list[at] = make_uint2(si * 32768u + e0, lo | ((e1 - e0) << 16));  /* first slot of the group; the range of its slots to look at */
}
__global__ __launch_bounds__(1024) void k_inf_ref_jump_last(uint32_t *ref, const uint2 *list, const uint32_t *count, uint32_t *open, uint32_t round)
{
    if (blockIdx.x >= *count) return;
    if (round && !open[round - 1u]) return;
    const uint2 g = list[blockIdx.x];
    uint32_t nr = ZI_REF_BYTE;
    if (threadIdx.x >= (g.y & 0xffffu) && threadIdx.x < (g.y >> 16)) {
        const uint32_t i = g.x + threadIdx.x;
        const uint32_t r = ref[i];
        nr = r;
        for (uint32_t h = 0; h < ZI_JUMP_HOPS && !(nr >> 30); h++) nr = __hip_atomic_load(ref + nr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nr != r) ref[i] = nr;
    }
    if (__ballot(!(nr >> 30)) && (threadIdx.x & 63u) == 0) open[round] = 1u;
}
__global__ __launch_bounds__(1024) void k_inf_ref_rest(const ZiSeg *segs, uint32_t *ref, uint32_t use, uint32_t chunk_size, uint64_t total)
{
    const uint32_t si = blockIdx.x >> 5, e = (blockIdx.x & 31u) * 1024u + threadIdx.x;
    const ZiSeg S = segs[si];
    const uint64_t ce = (uint64_t)(si / use + 1u) * chunk_size, chunk_end = ce < total ? ce : total;
    if (zi_in_last_window(S, e, chunk_end)) return;
    const uint64_t i = (uint64_t)si * 32768u + e;
    const uint32_t r = ref[i];
    if (!(r >> 30)) ref[i] = ref[r];                     /* a slot of a last window: resolved, and not written by this launch */
}
__global__ __launch_bounds__(1024) void k_inf_ref_write(const ZiSeg *segs, const uint32_t *ref, uint8_t *out8, const uint8_t *head)
{
    const uint32_t si = blockIdx.x >> 5, e = (blockIdx.x & 31u) * 1024u + threadIdx.x;
    const ZiSeg S = segs[si];
    const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
    if (e >= tl) return;
    const uint32_t r = ref[(uint64_t)si * 32768u + e];
    out8[S.out_off + S.out_len - tl + e] = (r & ZI_REF_HEAD) ? head[r & 32767u] : (uint8_t)r;
}
/* The last 32 KiB of (window in front of the shard ++ the shard's output) from the jumped references: what the
 * rank that holds the next shard needs as ITS window, long before the bulk of this shard's output is resolved. */
__global__ __launch_bounds__(1024) void k_inf_tail_window(const ZiSeg *segs, uint64_t nseg, const uint32_t *ref, const uint8_t *head,
                                                          uint64_t n_out, uint8_t *win)
{
    const uint32_t j = blockIdx.x * 1024u + threadIdx.x;
    if (j >= 32768u) return;
    const int64_t pos = (int64_t)n_out - 32768 + (int64_t)j;
    uint8_t b = 0;
    if (pos < 0) b = head ? head[32768 + pos] : (uint8_t)0;
    else {
        uint64_t si = nseg - 1;
        if (si > 0 && (uint64_t)pos < segs[si].out_off) si--;
        const ZiSeg S = segs[si];
        const uint64_t tl = S.out_len < 32768 ? S.out_len : 32768;
        const uint32_t r = ref[si * 32768u + ((uint64_t)pos - (S.out_off + S.out_len - tl))];
        b = (r & ZI_REF_HEAD) ? (head ? head[r & 32767u] : (uint8_t)0) : (uint8_t)r;
    }
    win[j] = b;
}

/* Everything in front of the tails: 16 elements per thread (two 16-byte loads, one 16-byte store; elements that are all
 * plain bytes, which is nearly all of them behind a segment's first 32 KiB, are packed four to an instruction). */
/* chunk_size != 0: segment tables of `parts` entries per chunk, the parts of a chunk in stream order with the unused
 * entries (empty, placed at the chunk's end) behind them */
#define ZI_FINAL_PER_THREAD 16u
__global__ __launch_bounds__(256) void k_inf_final(const ZiSeg *segs, uint64_t nseg, uint32_t chunk_size, uint32_t parts, const uint16_t *out16,
                                                   uint8_t *out8, const uint8_t *head, uint64_t total)
{
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * ZI_FINAL_PER_THREAD;
    if (p0 >= total) return;
    uint64_t si;
    if (chunk_size) {
        si = p0 / chunk_size * parts;
        for (uint32_t k = 1; k < parts && segs[si + 1].out_off <= p0; k++) si++;
    } else {
        uint64_t lo = 0, hi = nseg - 1;
        while (lo < hi) { uint64_t mid = (lo + hi + 1) >> 1; if (segs[mid].out_off <= p0) lo = mid; else hi = mid - 1; }
        si = lo;
    }
    ZiSeg S = segs[si];
    uint64_t seg_end = S.out_off + S.out_len;
    uint64_t tail0 = seg_end - (S.out_len < 32768 ? S.out_len : 32768);
    if (p0 + ZI_FINAL_PER_THREAD <= tail0) {
        const uint4 qa = *(const uint4 *)(out16 + p0), qb = *(const uint4 *)(out16 + p0 + 8);
        const uint32_t w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        uint32_t b[4];
        if (!((qa.x | qa.y | qa.z | qa.w | qb.x | qb.y | qb.z | qb.w) & 0xff00ff00u)) {
#pragma unroll
            for (int i = 0; i < 4; i++) b[i] = __builtin_amdgcn_perm(w[2 * i + 1], w[2 * i], 0x06040200u);     /* the low bytes of four elements */
        } else {
            b[0] = b[1] = b[2] = b[3] = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint32_t v = (w[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
                if (v >= ZI_MARK) v = zi_window_byte_written(S, v, out8, head);
                b[i >> 2] |= v << ((i & 3) * 8);
            }
        }
        *(uint4 *)(out8 + p0) = make_uint4(b[0], b[1], b[2], b[3]);
        return;
    }
    for (uint32_t i = 0; i < ZI_FINAL_PER_THREAD; i++) {
        const uint64_t p = p0 + i;
        if (p >= total) break;
        while (p >= seg_end && si + 1 < nseg) {
            si++; S = segs[si]; seg_end = S.out_off + S.out_len; tail0 = seg_end - (S.out_len < 32768 ? S.out_len : 32768);
        }
        if (p < S.out_off || p >= tail0) continue;      /* tails are already written */
        uint32_t v = out16[p];
        if (v >= ZI_MARK) v = zi_window_byte_written(S, v, out8, head);
        out8[p] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------------ host side */
struct ZiState {
    ZiSeg *d_segs = nullptr; uint64_t seg_cap = 0, mlist_cap = 0;
    uint16_t *d_stage = nullptr; uint64_t stage_cap = 0;     /* elements */
    uint32_t *d_marks = nullptr; uint32_t *d_mlist = nullptr;
    unsigned long long *d_res = nullptr;
    uint2 *d_flist = nullptr; uint32_t flist_cap = 0;        /* groups of slots in the chunks' last windows, and their count behind them */
    uint32_t last_status = 0; int64_t last_seg = -1;
    uint64_t *d_scan = nullptr; uint32_t scan_cap = 0;          /* zamd_find_sync_dev's result list */
    uint32_t *d_open = nullptr;                                 /* 64 words: which jump rounds left references behind */
    uint32_t *d_order = nullptr;                                /* seg_cap words: k_inf_order's list */
    uint64_t *d_bmarks = nullptr; uint64_t bmarks_cap = 0;      /* block marks of a counting pass, a row per segment */
    unsigned long long *h_bmarks = nullptr; uint32_t bmark_step = 0;     /* where the next counting pass is to leave them (zamd_inflate_block_marks) */
    /* a sharded inflate between its begin and finish calls */
    bool pending = false; bool pend_has_head = false; uint64_t pend_nseg = 0, pend_out_bytes = 0; uint32_t pend_chunk = 0; void *pend_out = nullptr;
};

static void zi_free(zamd_engine *e)
{
    ZiState *z = e->inf;
    if (!z) return;
    hipFree(z->d_segs); hipFree(z->d_stage); hipFree(z->d_marks); hipFree(z->d_mlist); hipFree(z->d_res); hipFree(z->d_flist); hipFree(z->d_scan); hipFree(z->d_bmarks); hipFree(z->d_open); hipFree(z->d_order);
    delete z; e->inf = nullptr;
}

static int zi_ensure(zamd_engine *e, uint64_t nseg, uint64_t stage_elems, bool resolve)
{
    if (!e->inf) { e->inf = new (std::nothrow) ZiState(); if (!e->inf) return ZAMD_MEM_ERROR; }
    ZiState *z = e->inf;
    /* every inflate entry point comes through here before it touches d_segs, d_mlist or d_stage: a sharded inflate that
     * was begun and not finished ends here (its tail and finish calls then report ZAMD_STREAM_ERROR instead of resolving
     * against buffers that belong to another call) */
    z->pending = false;
    if (!z->d_res) HIPCHK(hipMalloc((void **)&z->d_res, 16));
    if (!z->d_open) HIPCHK(hipMalloc((void **)&z->d_open, 64 * 4));
    if (z->seg_cap < nseg) {
        hipFree(z->d_segs); hipFree(z->d_marks); hipFree(z->d_order); z->d_segs = nullptr; z->d_marks = nullptr; z->d_order = nullptr; z->seg_cap = 0;
        uint64_t want = nseg + nseg / 4 + 16;
        if (hipMalloc((void **)&z->d_segs, want * sizeof(ZiSeg)) != hipSuccess) return ZAMD_MEM_ERROR;
        if (hipMalloc((void **)&z->d_marks, want * 4) != hipSuccess) return ZAMD_MEM_ERROR;
        if (hipMalloc((void **)&z->d_order, want * 4) != hipSuccess) return ZAMD_MEM_ERROR;
        z->seg_cap = want;
    }
    if (resolve && z->mlist_cap < nseg) {     /* marker lists: 32 Ki entries per segment */
        hipFree(z->d_mlist); z->d_mlist = nullptr; z->mlist_cap = 0;
        if (hipMalloc((void **)&z->d_mlist, nseg * 32768ull * 4) != hipSuccess) return ZAMD_MEM_ERROR;
        z->mlist_cap = nseg;
    }
    if (z->stage_cap < stage_elems) {
        hipFree(z->d_stage); z->d_stage = nullptr; z->stage_cap = 0;
        uint64_t want = stage_elems + 64;
        if (hipMalloc((void **)&z->d_stage, want * 2) != hipSuccess) return ZAMD_MEM_ERROR;
        z->stage_cap = want;
    }
    return ZAMD_OK;
}

/* decode (+ resolve when d_out != NULL) the segments already in z->d_segs */
static int zi_finish(zamd_engine *e, const void *d_head);
static int zi_run(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, uint64_t nseg, uint32_t chunk_size,
                  const void *d_head, void *d_out, uint64_t out_bytes, int64_t *bad_seg, bool regular_tails, bool defer = false, uint32_t parts = 1,
                  uint64_t *bmarks = nullptr, uint32_t mark_step = 0)
{
    ZiState *z = e->inf;
    hipStream_t s = e->st[0];
    int slot;
    e->n_tm = 0;
    HIPCHK(hipMemsetAsync(z->d_res, 0xff, 8, s));
    /* literal pairs in the speculative decode: 0 never, 1 always, 2 where a segment's tokens are mostly literals (tests and measurements set it) */
    static const uint32_t pair_mode = getenv("ZAMD_INF_PAIRS") ? (uint32_t)atoi(getenv("ZAMD_INF_PAIRS")) : 2u;
    /* more segments than wavefront slots, their input lengths known: the long ones first (ZAMD_INF_ORDER=0: as they come) */
    static const bool order_on = !(getenv("ZAMD_INF_ORDER") && atoi(getenv("ZAMD_INF_ORDER")) == 0);
    const uint32_t *order = nullptr;
    if (d_out && order_on && nseg > ZI_SPLIT_CHUNKS && nseg < (1ull << 31)) {
        hipLaunchKernelGGL(k_inf_order, dim3(1), dim3(1024), 0, s, z->d_segs, (uint32_t)nseg, z->d_order);
        order = z->d_order;
    }
    tm_begin(e, s, 7, &slot);
    if (d_out) hipLaunchKernelGGL((k_inf_decode<true>), dim3((uint32_t)nseg), dim3(64), 0, s, (const uint8_t *)d_comp, comp_bytes, z->d_segs, (uint32_t)nseg, z->d_stage,
                                  (uint64_t *)nullptr, 0u, pair_mode, order);
    else hipLaunchKernelGGL((k_inf_decode<false>), dim3((uint32_t)nseg), dim3(64), 0, s, (const uint8_t *)d_comp, comp_bytes, z->d_segs, (uint32_t)nseg, (uint16_t *)nullptr,
                            bmarks, mark_step, pair_mode, (const uint32_t *)nullptr);
    tm_end(e, s, slot);
    hipLaunchKernelGGL(k_inf_status, dim3(1), dim3(1024), 0, s, z->d_segs, nseg, z->d_res);
    unsigned long long bad = ~0ull;
    HIPCHK(hipMemcpyAsync(&bad, z->d_res, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    z->last_status = 0; z->last_seg = -1;
    int rc = ZAMD_OK;
    if (bad != ~0ull) {
        ZiSeg hs;
        HIPCHK(hipMemcpy(&hs, z->d_segs + bad, sizeof hs, hipMemcpyDeviceToHost));
        z->last_status = hs.status; z->last_seg = (int64_t)bad;
        if (bad_seg) *bad_seg = (int64_t)bad;
        rc = ZAMD_DATA_ERROR;
    }
    if (d_out && rc == ZAMD_OK && out_bytes) {
        tm_begin(e, s, 8, &slot);
        if (regular_tails && nseg <= 32768 && (defer || !getenv("ZAMD_INF_SERIAL_TAILS"))) {
            const uint64_t nref = nseg * 32768ull;
            hipLaunchKernelGGL(k_inf_ref_init, dim3((uint32_t)nseg * 32u), dim3(1024), 0, s, z->d_segs, z->d_stage, defer ? (const uint8_t *)nullptr : (const uint8_t *)d_head, z->d_mlist);
            static const bool flat_only = getenv("ZAMD_INF_FLAT_RESOLVE") != nullptr;      /* tests and measurements */
            if (parts > 1 && chunk_size >= 32768 && !defer && !flat_only) {
                const uint32_t nchunks = (uint32_t)(nseg / parts);
                for (uint32_t k = 1; k < parts; k++)
                    hipLaunchKernelGGL(k_inf_ref_local, dim3(nchunks * 32u), dim3(1024), 0, s, z->d_mlist, parts, k);
                /* a chunk's last 32 KiB lie in 32 groups of 1 024 slots, plus one for every segment border they cross */
                const uint32_t cap = nchunks * (33u + parts);
                if (z->flist_cap < cap + 1u) {
                    hipFree(z->d_flist); z->d_flist = nullptr; z->flist_cap = 0;
                    if (hipMalloc((void **)&z->d_flist, ((uint64_t)cap + 1u) * sizeof(uint2)) != hipSuccess) return ZAMD_MEM_ERROR;
                    z->flist_cap = cap + 1u;
                }
                uint32_t *const d_count = (uint32_t *)(z->d_flist + cap);
                HIPCHK(hipMemsetAsync(d_count, 0, 4, s));
                const uint64_t ngroups = nseg * 32ull;
                hipLaunchKernelGGL(k_inf_ref_last_list, dim3((uint32_t)((ngroups + 255) / 256)), dim3(256), 0, s, z->d_segs, ngroups, parts, chunk_size, out_bytes,
                                   z->d_flist, cap, d_count);
                uint32_t rounds = 1;
                while ((1ull << (ZI_JUMP_SHIFT * rounds)) < nchunks) rounds++;
                HIPCHK(hipMemsetAsync(z->d_open, 0, 64 * 4, s));
                for (uint32_t r = 0; r <= rounds; r++)
                    hipLaunchKernelGGL(k_inf_ref_jump_last, dim3(cap), dim3(1024), 0, s, z->d_mlist, z->d_flist, d_count, z->d_open, r);
                hipLaunchKernelGGL(k_inf_ref_rest, dim3((uint32_t)nseg * 32u), dim3(1024), 0, s, z->d_segs, z->d_mlist, parts, chunk_size, out_bytes);
            } else {
                uint32_t rounds = 1;
                while ((1ull << (ZI_JUMP_SHIFT * rounds)) < nseg) rounds++;
                HIPCHK(hipMemsetAsync(z->d_open, 0, 64 * 4, s));
                for (uint32_t r = 0; r <= rounds; r++)
                    hipLaunchKernelGGL(k_inf_ref_jump, dim3((uint32_t)((nref + 255) / 256)), dim3(256), 0, s, z->d_mlist, nref, z->d_open, r);
            }
            if (defer) {
                tm_end(e, s, slot);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(s));
                z->pending = true; z->pend_nseg = nseg; z->pend_out_bytes = out_bytes; z->pend_chunk = chunk_size; z->pend_out = d_out;
                return ZAMD_OK;
            }
            hipLaunchKernelGGL(k_inf_ref_write, dim3((uint32_t)nseg * 32u), dim3(1024), 0, s, z->d_segs, z->d_mlist, (uint8_t *)d_out, (const uint8_t *)d_head);
        } else {
            if (defer) return ZAMD_STREAM_ERROR;
            HIPCHK(hipMemsetAsync(z->d_marks, 0, nseg * 4, s));
            hipLaunchKernelGGL(k_inf_tail_pre, dim3((uint32_t)nseg * 32u), dim3(1024), 0, s, z->d_segs, z->d_stage, (uint8_t *)d_out, z->d_marks, z->d_mlist);
            hipLaunchKernelGGL(k_inf_tail_serial, dim3(1), dim3(1024), 0, s, z->d_segs, nseg, (uint8_t *)d_out, z->d_marks, z->d_mlist, (const uint8_t *)d_head);
        }
        uint64_t groups = (out_bytes + 256u * ZI_FINAL_PER_THREAD - 1) / (256u * ZI_FINAL_PER_THREAD);
        hipLaunchKernelGGL(k_inf_final, dim3((uint32_t)groups), dim3(256), 0, s, z->d_segs, nseg, chunk_size, parts, z->d_stage, (uint8_t *)d_out, (const uint8_t *)d_head, out_bytes);
        tm_end(e, s, slot);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    for (int k2 = 0; k2 < 12; k2++) { e->t_ms[k2] = 0; e->t_n[k2] = 0; }
    for (int i = 0; i < e->n_tm; i++) { float ms = 0; if (hipEventElapsedTime(&ms, e->tm[i].a, e->tm[i].b) == hipSuccess) { e->t_ms[e->tm[i].which] += ms; e->t_n[e->tm[i].which]++; } }
    return rc;
}

extern "C" int zamd_inflate_shard_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                      uint32_t chunk_size, const void *d_head_window, void *d_out, uint64_t n_out, int64_t *bad_chunk)
{
    if (!e || !d_comp || !d_chunk_off || !d_out || chunk_size == 0 || (chunk_size & 7u)) return ZAMD_STREAM_ERROR;
    if (((uintptr_t)d_comp & 3u) || ((uintptr_t)d_out & 15u)) return ZAMD_STREAM_ERROR;
    if (n_chunks == 0 || n_chunks > 0x7fffffffull / 32) return ZAMD_STREAM_ERROR;
    if ((n_out + chunk_size - 1) / chunk_size > n_chunks) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    int rc = zi_ensure(e, n_chunks, n_chunks * (uint64_t)chunk_size, true);
    if (rc) return rc;
    hipLaunchKernelGGL(k_inf_setup, dim3((uint32_t)((n_chunks + 255) / 256)), dim3(256), 0, e->st[0], d_chunk_off, n_chunks, chunk_size, n_out,
                       d_head_window ? 32768u : 0u, e->inf->d_segs);
    if (getenv("ZAMD_INF_COUNTONLY")) d_out = nullptr;      /* diagnostic: decode without writing */
    return zi_run(e, d_comp, comp_bytes, n_chunks, chunk_size, d_head_window, d_out, n_out, bad_chunk, chunk_size >= 32768);
}

/* Segment table of a chunk-indexed stream whose chunks come with split points (block boundaries inside the chunk, as
 * the deflate side records them): `parts` entries per chunk, the parts of the chunk in stream order, unused entries
 * empty at the chunk's end.  A split record is (bit offset from the chunk's first byte) << 32 | offset of the
 * block's first byte in the chunk's output; records must ascend in both halves and stay inside the chunk, and the
 * first one that does not ends the chunk's list (the rest of the chunk is then one part). */
/* `use` (<= parts): how many parts a chunk is cut into at most; of the chunk's parts - 1 records those nearest to the
 * multiples of 1 / use are taken. */
__global__ void k_inf_setup_parts(const uint64_t *chunk_off, uint64_t nchunks, uint32_t chunk_size, uint64_t n_out, uint64_t out_base,
                                  uint32_t head_avail, const uint64_t *splits, uint32_t parts, uint32_t use, ZiSeg *segs)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    const uint64_t bit0 = chunk_off[i] * 8ull, bit1 = chunk_off[i + 1] * 8ull, o0 = i * (uint64_t)chunk_size;
    const uint64_t rest = n_out > o0 ? n_out - o0 : 0, clen = rest < chunk_size ? rest : chunk_size;
    ZiSeg *out = segs + i * use;
    uint32_t idx = 0;
    uint64_t pbit = bit0, pout = 0;
    auto put = [&](uint64_t b0, uint64_t b1, uint64_t off, uint64_t cap) {
        ZiSeg s;
        s.in_bit = b0; s.in_end_bit = b1; s.out_off = o0 + off; s.out_cap = cap;
        const uint64_t before = out_base + s.out_off + head_avail;
        s.win_avail = (uint32_t)(before < 32768 ? before : 32768);
        s.mode = ZI_MODE_EXACT;
        s.out_len = 0; s.end_bit = 0; s.status = 0xffffffffu; s.final_seen = 0; s.stop_bit = 0;
        out[idx++] = s;
    };
    for (uint32_t j = 1; j < use; j++) {
        const uint32_t k = j * parts / use - 1u;
        const uint64_t rec = splits[i * (parts - 1u) + k];
        const uint64_t bit = bit0 + (rec >> 32), off = rec & 0xffffffffull;
        if (rec == 0 || off <= pout || off >= clen || bit <= pbit || bit >= bit1) break;
        put(pbit, bit, pout, off - pout);
        pbit = bit; pout = off;
    }
    put(pbit, bit1, pout, clen - pout);
    while (idx < use) put(bit1, bit1, clen, 0);
}

extern "C" int zamd_inflate_split_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                      uint32_t chunk_size, const uint64_t *d_splits, uint32_t parts, const void *d_head_window,
                                      void *d_out, uint64_t n_out, int64_t *bad_chunk)
{
    if (!d_splits || parts < 2)
        return zamd_inflate_shard_dev(e, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size, d_head_window, d_out, n_out, bad_chunk);
    if (!e || !d_comp || !d_chunk_off || !d_out || chunk_size == 0 || (chunk_size & 15u) || parts > ZAMD_MAX_PARTS) return ZAMD_STREAM_ERROR;
    if (((uintptr_t)d_comp & 3u) || ((uintptr_t)d_out & 15u)) return ZAMD_STREAM_ERROR;
    if (n_chunks == 0 || n_chunks > 0x7fffffffull / 32 / parts) return ZAMD_STREAM_ERROR;
    if ((n_out + chunk_size - 1) / chunk_size > n_chunks) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    /* How many of the parts to use: segments beyond what fills the GPU a few times over gain nothing in the decode (it
     * runs at the rate the whole GPU decodes at by then) and cost their 32 KiB each in the resolve step.  Measured on 1 MiB
     * chunks: 8 192 chunks are best left whole (in two parts 131 ms against 117, in three 120), 4 096 cut in five (1 to 3 %
     * better than in three on three corpora; six and more worse again), 512 in sixteen. */
    static const uint64_t seg_limit = getenv("ZAMD_INF_SPLIT_SEGMENTS") ? strtoull(getenv("ZAMD_INF_SPLIT_SEGMENTS"), nullptr, 10) : ZI_SPLIT_SEGMENTS;
    uint32_t use = parts;
    while (use > 1 && n_chunks * use > seg_limit) use--;
    if (n_chunks > ZI_SPLIT_CHUNKS && !getenv("ZAMD_INF_SPLIT_SEGMENTS")) use = 1;
    if (use < 2) return zamd_inflate_shard_dev(e, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size, d_head_window, d_out, n_out, bad_chunk);
    /* the references of the resolve step are 30 bits wide: 32 768 segments to a round, each round on top of the output
     * of the one before it */
    const uint64_t per_round = 32768u / use;
    const uint64_t first = n_chunks < per_round ? n_chunks : per_round;
    int rc = zi_ensure(e, first * use, first * (uint64_t)chunk_size, true);
    if (rc) return rc;
    double t_dec = 0, t_res = 0; int n_dec = 0, n_res = 0;
    for (uint64_t c0 = 0; c0 < n_chunks; c0 += per_round) {
        const uint64_t nc = n_chunks - c0 < per_round ? n_chunks - c0 : per_round;
        const uint64_t o0 = c0 * (uint64_t)chunk_size;
        const uint64_t out_here = n_out > o0 ? n_out - o0 : 0;
        const uint8_t *head = c0 ? (const uint8_t *)d_out + o0 - 32768 : (const uint8_t *)d_head_window;
        hipLaunchKernelGGL(k_inf_setup_parts, dim3((uint32_t)((nc + 255) / 256)), dim3(256), 0, e->st[0], d_chunk_off + c0, nc, chunk_size, out_here, o0,
                           d_head_window ? 32768u : 0u, d_splits + c0 * (parts - 1u), parts, use, e->inf->d_segs);
        int64_t bad = -1;
        rc = zi_run(e, d_comp, comp_bytes, nc * use, chunk_size, head, (uint8_t *)d_out + o0, out_here < nc * (uint64_t)chunk_size ? out_here : nc * (uint64_t)chunk_size,
                    &bad, true, false, use);
        if (rc == ZAMD_DATA_ERROR) { e->inf->last_seg = (int64_t)c0 + bad / (int64_t)use; if (bad_chunk) *bad_chunk = e->inf->last_seg; }
        if (rc) return rc;
        t_dec += e->t_ms[7]; n_dec += e->t_n[7]; t_res += e->t_ms[8]; n_res += e->t_n[8];
    }
    e->t_ms[7] = t_dec; e->t_n[7] = n_dec; e->t_ms[8] = t_res; e->t_n[8] = n_res;
    return ZAMD_OK;
}

/* Second half of a deferred zi_run: the window in front of the shard has arrived. */
static int zi_finish(zamd_engine *e, const void *d_head)
{
    ZiState *z = e->inf;
    hipStream_t s = e->st[0];
    const uint64_t nseg = z->pend_nseg;
    int slot;
    tm_begin(e, s, 8, &slot);
    hipLaunchKernelGGL(k_inf_ref_write, dim3((uint32_t)nseg * 32u), dim3(1024), 0, s, z->d_segs, z->d_mlist, (uint8_t *)z->pend_out, (const uint8_t *)d_head);
    const uint64_t groups = (z->pend_out_bytes + 256u * ZI_FINAL_PER_THREAD - 1) / (256u * ZI_FINAL_PER_THREAD);
    hipLaunchKernelGGL(k_inf_final, dim3((uint32_t)groups), dim3(256), 0, s, z->d_segs, nseg, z->pend_chunk, 1u, z->d_stage, (uint8_t *)z->pend_out, (const uint8_t *)d_head, z->pend_out_bytes);
    tm_end(e, s, slot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    z->pending = false;
    for (int k2 = 0; k2 < 12; k2++) { e->t_ms[k2] = 0; e->t_n[k2] = 0; }
    for (int i = 0; i < e->n_tm; i++) { float ms = 0; if (hipEventElapsedTime(&ms, e->tm[i].a, e->tm[i].b) == hipSuccess) { e->t_ms[e->tm[i].which] += ms; e->t_n[e->tm[i].which]++; } }
    return ZAMD_OK;
}

extern "C" int zamd_inflate_shard_begin_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                            uint32_t chunk_size, int has_head_window, void *d_out, uint64_t n_out, int64_t *bad_chunk)
{
    if (!e || !d_comp || !d_chunk_off || !d_out || chunk_size < 32768 || (chunk_size & 7u)) return ZAMD_STREAM_ERROR;
    if (((uintptr_t)d_comp & 3u) || ((uintptr_t)d_out & 15u)) return ZAMD_STREAM_ERROR;
    if (n_chunks == 0 || n_chunks >= 32768 || n_out == 0) return ZAMD_STREAM_ERROR;
    if ((n_out + chunk_size - 1) / chunk_size != n_chunks) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    int rc = zi_ensure(e, n_chunks, n_chunks * (uint64_t)chunk_size, true);
    if (rc) return rc;
    e->inf->pending = false; e->inf->pend_has_head = has_head_window != 0;
    hipLaunchKernelGGL(k_inf_setup, dim3((uint32_t)((n_chunks + 255) / 256)), dim3(256), 0, e->st[0], d_chunk_off, n_chunks, chunk_size, n_out,
                       has_head_window ? 32768u : 0u, e->inf->d_segs);
    return zi_run(e, d_comp, comp_bytes, n_chunks, chunk_size, nullptr, d_out, n_out, bad_chunk, true, true);
}

extern "C" int zamd_inflate_shard_tail_dev(zamd_engine *e, const void *d_head_window, void *d_tail_window)
{
    if (!e || !e->inf || !e->inf->pending || !d_tail_window) return ZAMD_STREAM_ERROR;
    ZiState *z = e->inf;
    if (z->pend_has_head && !d_head_window) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_inf_tail_window, dim3(32), dim3(1024), 0, e->st[0], z->d_segs, z->pend_nseg, z->d_mlist,
                       z->pend_has_head ? (const uint8_t *)d_head_window : (const uint8_t *)nullptr, z->pend_out_bytes, (uint8_t *)d_tail_window);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->st[0]));
    return ZAMD_OK;
}

extern "C" int zamd_inflate_shard_finish_dev(zamd_engine *e, const void *d_head_window)
{
    if (!e || !e->inf || !e->inf->pending) return ZAMD_STREAM_ERROR;
    if (e->inf->pend_has_head && !d_head_window) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    return zi_finish(e, e->inf->pend_has_head ? d_head_window : nullptr);
}

extern "C" int zamd_inflate_dev(zamd_engine *e, const void *d_comp, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                uint32_t chunk_size, void *d_out, uint64_t n_out, int64_t *bad_chunk)
{
    if (!e || !d_chunk_off || n_chunks == 0) return ZAMD_STREAM_ERROR;
    uint64_t comp_bytes = 0;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpy(&comp_bytes, d_chunk_off + n_chunks, 8, hipMemcpyDeviceToHost));
    return zamd_inflate_shard_dev(e, d_comp, comp_bytes, d_chunk_off, n_chunks, chunk_size, nullptr, d_out, n_out, bad_chunk);
}

extern "C" int zamd_inflate_segments_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, zamd_inflate_seg *segs, uint64_t nseg,
                                         const void *d_head_window, void *d_out, uint64_t out_bytes)
{
    static_assert(sizeof(zamd_inflate_seg) == sizeof(ZiSeg), "segment layouts must agree");
    if (!e || !d_comp || !segs || nseg == 0 || nseg > 0x7fffffffull / 32) return ZAMD_STREAM_ERROR;
    if (((uintptr_t)d_comp & 3u) || ((uintptr_t)d_out & 15u)) return ZAMD_STREAM_ERROR;
    uint64_t need = 0;
    for (uint64_t i = 0; i < nseg; i++) {
        if (segs[i].in_end_bit > comp_bytes * 8 || segs[i].in_bit > segs[i].in_end_bit || segs[i].win_avail > 32768) return ZAMD_STREAM_ERROR;
        if (d_out) {
            if (segs[i].out_off + segs[i].out_cap > out_bytes) return ZAMD_STREAM_ERROR;
            if (i && segs[i].out_off < segs[i - 1].out_off + segs[i - 1].out_cap) return ZAMD_STREAM_ERROR;
            if (segs[i].win_avail > segs[i].out_off && !d_head_window) return ZAMD_STREAM_ERROR;
            need = segs[i].out_off + segs[i].out_cap;
        }
        segs[i].status = 0xffffffffu;
    }
    HIPCHK(hipSetDevice(e->device));
    int rc = zi_ensure(e, nseg, need, d_out != nullptr);
    if (rc) return rc;
    /* a request for block marks holds for this call only, and only a counting pass serves it */
    ZiState *const z = e->inf;
    unsigned long long *const h_bmarks = d_out ? nullptr : z->h_bmarks;
    const uint32_t mark_step = z->bmark_step;
    z->h_bmarks = nullptr; z->bmark_step = 0;
    if (h_bmarks && z->bmarks_cap < nseg) {
        hipFree(z->d_bmarks); z->d_bmarks = nullptr; z->bmarks_cap = 0;
        const uint64_t want = nseg + nseg / 4 + 16;
        if (hipMalloc((void **)&z->d_bmarks, want * (ZI_MARKS + 1u) * 8u) != hipSuccess) return ZAMD_MEM_ERROR;
        z->bmarks_cap = want;
    }
    HIPCHK(hipMemcpy(e->inf->d_segs, segs, nseg * sizeof(ZiSeg), hipMemcpyHostToDevice));
    /* every segment is reported, so a failing one is not an error of the call */
    int64_t bad = -1;
    bool regular = segs[0].out_off == 0;   /* and the segments back to back, their lengths known beforehand */
    for (uint64_t i = 0; i + 1 < nseg && regular; i++)
        regular = segs[i].out_off + segs[i].out_cap == segs[i + 1].out_off && (segs[i].mode & ZAMD_INF_EXACT);
    rc = zi_run(e, d_comp, comp_bytes, nseg, 0, d_head_window, d_out, need, &bad, regular, false, 1, h_bmarks ? z->d_bmarks : nullptr, mark_step);
    if (rc != ZAMD_OK && rc != ZAMD_DATA_ERROR) return rc;
    HIPCHK(hipMemcpy(segs, e->inf->d_segs, nseg * sizeof(ZiSeg), hipMemcpyDeviceToHost));
    if (h_bmarks) HIPCHK(hipMemcpy(h_bmarks, z->d_bmarks, nseg * (ZI_MARKS + 1u) * 8u, hipMemcpyDeviceToHost));
    return ZAMD_OK;
}

extern "C" int zamd_inflate_block_marks(zamd_engine *e, unsigned long long *h_marks, uint32_t step_bytes)
{
    static_assert(ZAMD_INF_MARKS == ZI_MARKS + 1u, "a row of block marks has one entry more than a segment can report");
    if (!e || (h_marks && step_bytes == 0)) return ZAMD_STREAM_ERROR;
    if (!e->inf) { e->inf = new (std::nothrow) ZiState(); if (!e->inf) return ZAMD_MEM_ERROR; }
    e->inf->h_bmarks = h_marks; e->inf->bmark_step = h_marks ? step_bytes : 0u;
    return ZAMD_OK;
}

/* Positions just behind every 00 00 FF FF in d_in[from, n): where a block may start after a sync marker
 * (the pattern inflateSync looks for, inflate.c:1354-1375). */
#define ZI_SCAN_PER_THREAD 16u
__global__ __launch_bounds__(256) void k_inf_scan(const uint8_t *in, uint64_t from, uint64_t n, uint64_t *list, uint32_t cap, uint32_t *count)
{
    const uint64_t base = from + ((uint64_t)blockIdx.x * 256u + threadIdx.x) * ZI_SCAN_PER_THREAD;
    if (base + 4 > n) return;
    uint32_t w = (uint32_t)in[base] | ((uint32_t)in[base + 1] << 8) | ((uint32_t)in[base + 2] << 16);
    for (uint32_t k = 0; k < ZI_SCAN_PER_THREAD && base + k + 4 <= n; k++) {
        w |= (uint32_t)in[base + k + 3] << 24;
        if (w == 0xffff0000u) { uint32_t i = atomicAdd(count, 1u); if (i < cap) list[i] = base + k + 4; }
        w >>= 8;
    }
}

extern "C" int zamd_find_sync_dev(zamd_engine *e, const void *d_in, uint64_t from, uint64_t n, uint64_t *h_list, uint32_t cap, uint32_t *count)
{
    if (!e || !d_in || !h_list || !count || from > n) return ZAMD_STREAM_ERROR;
    *count = 0;
    if (n - from < 4) return ZAMD_OK;
    HIPCHK(hipSetDevice(e->device));
    int rc = zi_ensure(e, 1, 0, false);
    if (rc) return rc;
    ZiState *z = e->inf;
    if (z->scan_cap < cap) {
        hipFree(z->d_scan); z->d_scan = nullptr; z->scan_cap = 0;
        if (hipMalloc((void **)&z->d_scan, (uint64_t)cap * 8) != hipSuccess) return ZAMD_MEM_ERROR;
        z->scan_cap = cap;
    }
    hipStream_t s = e->st[0];
    uint32_t *d_count = (uint32_t *)z->d_res;
    HIPCHK(hipMemsetAsync(d_count, 0, 4, s));
    const uint64_t threads = (n - from + ZI_SCAN_PER_THREAD - 1) / ZI_SCAN_PER_THREAD;
    hipLaunchKernelGGL(k_inf_scan, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, s, (const uint8_t *)d_in, from, n, z->d_scan, cap, d_count);
    HIPCHK(hipGetLastError());
    uint32_t c = 0;
    HIPCHK(hipMemcpyAsync(&c, d_count, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *count = c;                                     /* more than cap: the list is incomplete, the caller falls back */
    if (c && c <= cap) HIPCHK(hipMemcpy(h_list, z->d_scan, (uint64_t)c * 8, hipMemcpyDeviceToHost));
    return ZAMD_OK;
}

/* ------------------------------------------------------------------ block-start search
 * One workgroup per target; lane l of round r probes the bits target + 512 r + 8 l .. + 7.  A probe reads what inflate reads
 * between a block's first bit and its first symbol (inflate.c:894-1012) and applies inflate's checks, plus
 * two that hold for every encoder of interest: the block is not the last one, and its literal/length code is
 * complete.  The code-length code is decoded canonically from its length counts (no table): probes that get
 * that far are rare. */
__device__ __forceinline__ uint64_t zi_bits_at(const uint8_t *in, uint64_t nbytes, uint64_t bit)
{
    const uint64_t byte = bit >> 3;
    uint64_t v = 0;
    if (byte + 8 <= nbytes) __builtin_memcpy(&v, in + byte, 8);
    else for (uint32_t i = 0; i < 8 && byte + i < nbytes; i++) v |= (uint64_t)in[byte + i] << (8u * i);
    return v >> (bit & 7u);                      /* 57 bits or more */
}

/* The part of the probe that reads 74 bits at most: block type, counts, and a complete code-length code without
 * trailing zero lengths.  One position in about five hundred passes it on data that holds no block start. */
__device__ bool zi_probe_head(const uint8_t *in, uint64_t nbytes, uint64_t bit)
{
    const uint64_t end = nbytes * 8ull;
    if (bit + 17 + 12 > end) return false;
    uint64_t b = zi_bits_at(in, nbytes, bit);
    if ((b & 7u) != 4u) return false;
    const uint32_t nlen = ((uint32_t)(b >> 3) & 31u) + 257u, ndist = ((uint32_t)(b >> 8) & 31u) + 1u, ncode = ((uint32_t)(b >> 13) & 15u) + 4u;
    if (nlen > 286 || ndist > 30) return false;
    b = zi_bits_at(in, nbytes, bit + 17) & ((1ull << (3u * ncode)) - 1ull);
    const uint64_t M = 0x1249249249249249ull;
    const uint64_t a0 = b & M, a1 = (b >> 1) & M, a2 = (b >> 2) & M, n0 = a0 ^ M, n1 = a1 ^ M, n2 = a2 ^ M;
    const uint32_t kraft = 64u * __popcll(a0 & n1 & n2) + 32u * __popcll(n0 & a1 & n2) + 16u * __popcll(a0 & a1 & n2) + 8u * __popcll(n0 & n1 & a2) +
                           4u * __popcll(a0 & n1 & a2) + 2u * __popcll(n0 & a1 & a2) + __popcll(a0 & a1 & a2);
    if (kraft != 128u) return false;
    return ncode == 4 || ((b >> (3u * (ncode - 1u))) & 7u) != 0;
}

__device__ bool zi_probe_block(const uint8_t *in, uint64_t nbytes, uint64_t bit)
{
    const uint64_t end = nbytes * 8ull;
    if (bit + 17 + 12 > end) return false;
    uint64_t b = zi_bits_at(in, nbytes, bit);
    if ((b & 7u) != 4u) return false;                                  /* BFINAL 0, BTYPE 2 */
    const uint32_t nlen = ((uint32_t)(b >> 3) & 31u) + 257u, ndist = ((uint32_t)(b >> 8) & 31u) + 1u, ncode = ((uint32_t)(b >> 13) & 15u) + 4u;
    if (nlen > 286 || ndist > 30) return false;
    uint64_t pos = bit + 17;
    b = zi_bits_at(in, nbytes, pos) & ((1ull << (3u * ncode)) - 1ull);
    /* how many of the 3-bit lengths have each value, from the three bit planes: the code must be complete
     * (inflate wants that, inflate.c:929) — the test that ends nearly every probe that got this far */
    uint64_t cnt;
    {
        const uint64_t M = 0x1249249249249249ull;
        const uint64_t a0 = b & M, a1 = (b >> 1) & M, a2 = (b >> 2) & M, n0 = a0 ^ M, n1 = a1 ^ M, n2 = a2 ^ M;
        const uint32_t c1 = __popcll(a0 & n1 & n2), c2 = __popcll(n0 & a1 & n2), c3 = __popcll(a0 & a1 & n2), c4 = __popcll(n0 & n1 & a2),
                       c5 = __popcll(a0 & n1 & a2), c6 = __popcll(n0 & a1 & a2), c7 = __popcll(a0 & a1 & a2);
        if (64u * c1 + 32u * c2 + 16u * c3 + 8u * c4 + 4u * c5 + 2u * c6 + c7 != 128u) return false;
        cnt = ((uint64_t)c1 << 5) | ((uint64_t)c2 << 10) | ((uint64_t)c3 << 15) | ((uint64_t)c4 << 20) | ((uint64_t)c5 << 25) |
              ((uint64_t)c6 << 30) | ((uint64_t)c7 << 35);
    }
    /* an encoder sends no trailing zero lengths (trees.c:815-821, 842-848 send max_blindex + 1, max_code + 1 codes):
     * not a rule of the format, but a guess need not cover every stream */
    if (ncode > 4 && ((b >> (3u * (ncode - 1u))) & 7u) == 0) return false;
    uint64_t pk = 0;                                                   /* 3 bits per symbol */
    for (uint32_t i = 0; i < ncode; i++) pk |= ((b >> (3u * i)) & 7ull) << (3u * zi_order[i]);
    pos += 3ull * ncode;
    const uint32_t total = nlen + ndist;
    /* symbols in (length, symbol) order, 5 bits each: the order canonical codes are handed out in */
    uint64_t so0 = 0, so1 = 0;
    {
        uint32_t k = 0;
        for (uint32_t L = 1; L <= 7; L++)
            for (uint32_t sy = 0; sy < 19; sy++)
                if (((uint32_t)(pk >> (3u * sy)) & 7u) == L) {
                    if (k < 12) so0 |= (uint64_t)sy << (5u * k); else so1 |= (uint64_t)sy << (5u * (k - 12u));
                    k++;
                }
    }
    uint32_t idx = 0, prev = 0, sumL = 0, sumD = 0, nzD = 0, maxD = 0, last_l = 0, last_d = 0;
    bool eob = false;
    uint32_t have = 0;                                                 /* valid bits in b */
    while (idx < total) {
        if (have < 14) {
            if (pos + 14 > end) return false;
            b = zi_bits_at(in, nbytes, pos); have = 57;
        }
        /* canonical decode: codes of one length are consecutive, shorter codes come first */
        uint32_t code = 0, first = 0, index = 0, len = 1, sym = 99;
        for (; len <= 7; len++) {
            code |= (uint32_t)(b >> (len - 1)) & 1u;
            const uint32_t c = (uint32_t)(cnt >> (5u * len)) & 31u;
            if (code < first + c) {
                const uint32_t k = index + (code - first);
                sym = k < 12 ? (uint32_t)(so0 >> (5u * k)) & 31u : (uint32_t)(so1 >> (5u * (k - 12u))) & 31u;
                break;
            }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        if (sym == 99) return false;
        uint32_t used = len;
        b >>= len;
        uint32_t rep = 1, val = sym;
        if (sym == 16) { if (idx == 0) return false; rep = 3 + ((uint32_t)b & 3u); used += 2; b >>= 2; val = prev; }
        else if (sym == 17) { rep = 3 + ((uint32_t)b & 7u); used += 3; b >>= 3; val = 0; }
        else if (sym == 18) { rep = 11 + ((uint32_t)b & 127u); used += 7; b >>= 7; val = 0; }
        pos += used; have -= used;
        if (idx + rep > total) return false;
        prev = val;
        if (val) {
            for (uint32_t j = 0; j < rep; j++) {
                const uint32_t i = idx + j;
                if (i < nlen) { sumL += 32768u >> val; if (i == 256) eob = true; if (i == nlen - 1) last_l = val; }
                else { sumD += 32768u >> val; nzD++; if (val > maxD) maxD = val; if (i == total - 1) last_d = val; }
            }
            if (sumL > 32768u || sumD > 32768u) return false;          /* over-subscribed already */
        }
        idx += rep;
    }
    if (pos > end || !eob) return false;
    if (sumL != 32768u) return false;
    if ((nlen > 257 && last_l == 0) || (ndist > 1 && last_d == 0)) return false;
    return sumD == 32768u || (nzD <= 1 && maxD <= 1);
}

#define ZI_BSCAN_THREADS 1024
__global__ __launch_bounds__(ZI_BSCAN_THREADS) void k_inf_blockscan(const uint8_t *in, uint64_t nbytes, const uint64_t *targets, uint32_t ntargets,
                                                                   uint64_t window_bits, uint64_t *found)
{
    /* the 16 wavefronts of a workgroup take the rounds of one target in turn; a hit ends every round behind it */
    __shared__ unsigned long long s_hit;
    __shared__ unsigned long long s_q[ZI_BSCAN_THREADS / 64][128];
    const uint32_t t = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (t >= ntargets) return;
    if (threadIdx.x == 0) s_hit = ~0ull;
    __syncthreads();
    const uint64_t t0 = targets[t];
    /* A lane takes the eight bit positions of one byte (one load).  Positions that pass the short part of the probe
     * are queued per wavefront and get the long part — the decode of up to 316 code lengths — 64 at a time, one per
     * lane: done where they are found, one such position would hold up the 63 other lanes of its round. */
    unsigned long long *q = s_q[wave];
    uint32_t qn = 0;
    auto drain = [&](uint32_t n) {
        __threadfence_block();
        if (lane < n) {
            const unsigned long long pos = q[lane];
            if (zi_probe_block(in, nbytes, pos)) atomicMin(&s_hit, pos);
        }
        unsigned long long keep = 0;
        if (lane < qn - n) keep = q[n + lane];
        __threadfence_block();
        if (lane < qn - n) q[lane] = keep;
        qn -= n;
    };
    for (uint64_t r = 512ull * wave; r < window_bits; r += 512ull * (ZI_BSCAN_THREADS / 64)) {
        if (*(volatile unsigned long long *)&s_hit < t0 + r) break;
        const uint64_t off = r + 8ull * lane, bit0 = t0 + off;
        uint32_t pm = 0;
        if (off < window_bits) {
            const uint64_t w = zi_bits_at(in, nbytes, bit0);
            for (uint32_t s8 = 0; s8 < 8; s8++) {
                const uint32_t h = (uint32_t)(w >> s8);
                if ((h & 7u) == 4u && ((h >> 3) & 31u) <= 29u && ((h >> 8) & 31u) <= 29u && off + s8 < window_bits &&
                    zi_probe_head(in, nbytes, bit0 + s8)) pm |= 1u << s8;
            }
        }
        if (__ballot(pm != 0) == 0ull) continue;
        for (uint32_t s8 = 0; s8 < 8; s8++) {
            const bool has = (pm >> s8) & 1u;
            const unsigned long long m = __ballot(has);
            if (m == 0ull) continue;
            if (has) q[qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = bit0 + s8;
            qn += (uint32_t)__popcll(m);
            if (qn >= 64u) drain(64u);
        }
    }
    if (qn) drain(qn);
    __syncthreads();
    if (threadIdx.x == 0) found[t] = s_hit;
}

extern "C" int zamd_find_blocks_dev(zamd_engine *e, const void *d_in, uint64_t n, const uint64_t *h_targets_bit, uint32_t ntargets,
                                    uint64_t window_bits, uint64_t *h_found_bit)
{
    if (!e || !d_in || !h_targets_bit || !h_found_bit) return ZAMD_STREAM_ERROR;
    if (ntargets == 0) return ZAMD_OK;
    for (uint32_t i = 0; i < ntargets; i++) if (h_targets_bit[i] > n * 8ull) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    int rc = zi_ensure(e, 1, 0, false);
    if (rc) return rc;
    ZiState *z = e->inf;
    const uint32_t need = 2u * ntargets;
    if (z->scan_cap < need) {
        hipFree(z->d_scan); z->d_scan = nullptr; z->scan_cap = 0;
        if (hipMalloc((void **)&z->d_scan, (uint64_t)need * 8) != hipSuccess) return ZAMD_MEM_ERROR;
        z->scan_cap = need;
    }
    hipStream_t s = e->st[0];
    HIPCHK(hipMemcpyAsync(z->d_scan, h_targets_bit, (uint64_t)ntargets * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_inf_blockscan, dim3(ntargets), dim3(ZI_BSCAN_THREADS), 0, s, (const uint8_t *)d_in, n, z->d_scan, ntargets, window_bits, z->d_scan + ntargets);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h_found_bit, z->d_scan + ntargets, (uint64_t)ntargets * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ZAMD_OK;
}

extern "C" const char *zamd_inflate_message(uint32_t status) { return zi_message(status); }

extern "C" int zamd_inflate_last_error(zamd_engine *e, int64_t *segment, uint32_t *status)
{
    if (!e || !e->inf) return ZAMD_STREAM_ERROR;
    if (segment) *segment = e->inf->last_seg;
    if (status) *status = e->inf->last_status;
    return ZAMD_OK;
}
