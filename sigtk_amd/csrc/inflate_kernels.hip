// inflate_kernels.hip -- zlib (RFC 1950 / 1951) streams inflated on the GPU, one wavefront per stream (round 5).
//
// Why: BLOW5 files written by slow5lib compress every record as one zlib stream (slow5lib/src/slow5.c:2583-2598,
// slow5_press.c:77-98); the reference inflates them one at a time on one thread (slow5_get_next, src/cmain.c:118).  The
// drop-in CLI inflated them on a pool of host threads -- and that is what its steady-state rate was: 5 GB/s of inflated
// bytes on the 16 cores a GPU box gives a job (profiles/r05_k_cli_steady_before.json: 2.5 of 3.6 s for 1e10 samples), a
// third of a percent of what the kernels behind it take.  Here the records go to the GPU as they sit in the file, a
// wavefront inflates each into device memory, and the svb-zd decoder (svb_kernels.hip) reads the signal blob from there.
//
// One stream per wavefront.  DEFLATE is serial in its bit stream, so the symbol loop runs on wave-uniform values (bit
// buffer, positions, table entries: scalar registers) and the lanes work together where there is something to do together:
//   * the input is read 256 bytes at a time, one dword per lane (coalesced); the bit buffer takes its dwords with v_readlane;
//   * the newest 4 KB of output are a ring in LDS; completed 1 KB chunks leave for global memory as 16-byte stores, their
//     Adler-32 partial sums reduced across the wave on the way.  A match copies up to 64 bytes per step, lane i the byte
//     at distance dist - (i mod dist) (overlapping matches included) -- from the ring when it reaches back less than
//     3 838 bytes, else from the stream's own bytes in global memory, which have left the ring by then.  (The first
//     version kept the whole 32 KB window in LDS: four waves per compute unit, each alone on its SIMD and bound by the
//     latency of its own instruction chain -- 5 GB/s for the whole GPU, what 16 host cores do.  With 8 KB of LDS a
//     compute unit holds 19 streams.)
//   * Huffman tables are built in parallel (a lane per symbol: rank within its code length by ballots, bit-reversed
//     replication into a 10-bit / 8-bit look-up table); codes longer than the table walk the canonical first-code /
//     count arrays (puff.c's decode), which is rare.
// Everything a zlib inflate checks is checked: header, block types, stored-block length complement, over-subscribed code
// length sets, invalid codes, distances in front of the stream, truncated input, Adler-32.  A stream that fails leaves a
// non-zero status and the host decides (the CLI reports it as the reference does a slow5_get_next error).
#include "sgk_common.h"

namespace sgk {

constexpr int INF_WIN = 4096;    // the output ring in LDS: the newest bytes (a match that reaches further back reads global memory)
constexpr int INF_NEAR = INF_WIN - 258;   // a match of up to 258 bytes at up to this distance lies in the ring, and stays intact while it is copied
constexpr int INF_LB = 10;       // literal / length look-up table bits
constexpr int INF_DB = 8;        // distance look-up table bits
constexpr int INF_FLUSH = 1024;  // bytes per flush of the ring to global memory (64 lanes x 16)

// status codes (sgk_inflate's per-stream status)
enum {
    INF_OK = 0,
    INF_ERR_HEADER = 1,     // not a zlib stream (CMF / FLG), or a preset dictionary
    INF_ERR_BLOCK = 2,      // reserved block type / stored-block length check
    INF_ERR_LENGTHS = 3,    // invalid or over-subscribed code length set, no end-of-block code
    INF_ERR_CODE = 4,       // invalid literal / length or distance code
    INF_ERR_DISTANCE = 5,   // distance reaches in front of the stream
    INF_ERR_TRUNCATED = 6,  // input ends inside the stream
    INF_ERR_ADLER = 7,      // check value mismatch
    INF_ERR_ROOM = 8,       // the stream inflates to more than out_caps[r] bytes
};

struct InfArgs {
    const uint8_t *in;             // all streams
    const uint64_t *in_offsets;    // n: byte offset of stream r
    const uint32_t *in_lengths;    // n: its bytes
    uint8_t *out;                  // inflated bytes
    const uint64_t *out_offsets;   // n: 16-byte aligned offsets into out
    const uint32_t *out_caps;      // n: room for stream r
    uint32_t *out_lengths;         // n: bytes the stream inflates to
    uint32_t *status;              // n
    uint32_t n;
};

template <int TB, int NS>
struct InfTables {
    uint16_t fast[1 << TB];        // (symbol << 4) | length for codes of up to TB bits, 0: longer / invalid
    uint32_t count[16];            // codes per length
    uint16_t first[16];            // canonical first code of each length
    uint16_t offs[16];             // index of the first symbol of each length in sym[]
    uint32_t longw[16];            // per length: (first + count) << 16 | (offs - first) & 0xffff -- a code of that length
                                   // is `code` iff code < first + count, and its symbol sym[code + (offs - first)]
    uint16_t sym[NS];              // symbols ordered by (length, symbol)
};
struct InfLds {
    uint8_t win[INF_WIN];
    InfTables<INF_LB, 288> lit;
    union {
        InfTables<INF_DB, 32> dist;
        InfTables<7, 20> cl;       // the code length code: done with before the distance table is built
    };
    uint8_t lens[320];             // code lengths of the current block
};
static_assert(sizeof(InfLds) <= 8448, "19 streams per compute unit");

__constant__ const uint8_t INF_CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// ---- the bit reader: wave-uniform state, the input window one dword per lane
struct InfBits {
    const uint32_t *base;   // 4-byte aligned start of the stream's dwords
    uint32_t n_dw;          // dwords that may be read
    uint32_t end_bit;       // first bit behind the stream (from base)
    uint32_t win, nxt;      // per lane: dword (chunk * 64 + lane) of the current chunk and of the one behind it (in flight)
    uint32_t chunk;         // which 64-dword chunk `win` holds
    uint32_t next_dw;       // next dword to enter the bit buffer
    unsigned long long buf; // bits, LSB first
    uint32_t cnt;           // valid bits in buf

    __device__ __forceinline__ uint32_t dword(uint32_t k) {
        const uint32_t c = k >> 6;
        if (c != chunk) {
            if (c == chunk + 1u) win = nxt;   // (requested a chunk ago)
            else {
                const uint32_t i = c * 64u + (uint32_t)lane_id();
                win = i < n_dw ? base[i] : 0u;
            }
            chunk = c;
            const uint32_t j = (c + 1u) * 64u + (uint32_t)lane_id();
            nxt = j < n_dw ? base[j] : 0u;
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(k & 63u));
    }
    // dword k without moving the window (k in the current chunk or the one behind it, else 0: the caller stays within
    // four dwords of the bit buffer)
    __device__ __forceinline__ uint32_t peek_dword(uint32_t k) const {
        const uint32_t c = k >> 6;
        const uint32_t v = c == chunk ? win : nxt;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(k & 63u));
        return (c == chunk || c == chunk + 1u) ? x : 0u;
    }
    __device__ __forceinline__ void refill() {   // at least 32 valid bits afterwards
        if (cnt <= 32u) {
            buf |= (unsigned long long)dword(next_dw) << cnt;
            ++next_dw;
            cnt += 32u;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)buf & ((1u << n) - 1u); }   // n <= 31
    __device__ __forceinline__ void drop(uint32_t n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t get(uint32_t n) {   // n <= 16, after refill()
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    __device__ __forceinline__ uint32_t bit_pos() const { return next_dw * 32u - cnt; }
    __device__ __forceinline__ bool overrun() const { return bit_pos() > end_bit; }
    __device__ __forceinline__ void seek_bit(uint32_t bit) {
        next_dw = bit >> 5;
        buf = 0ull;
        cnt = 0u;
        refill();
        drop(bit & 31u);
        refill();
    }
    __device__ __forceinline__ void seek_byte(uint32_t byte) {
        next_dw = byte >> 2;
        buf = 0ull;
        cnt = 0u;
        refill();
        drop((byte & 3u) * 8u);
        refill();
    }
};

// ---- Huffman tables, built by the whole wave from lens[0 .. nsym) (in LDS)
// returns false for an over-subscribed set of code lengths
template <int TB, int NS>
__device__ bool inf_build(InfTables<TB, NS> *t, const uint8_t *lens, int nsym) {
    const int l = lane_id();
    if (l < 16) t->count[l] = 0u;
    for (int k = l; k < (1 << TB); k += 64) t->fast[k] = 0;
    __syncthreads();
    for (int s = l; s < nsym; s += 64) atomicAdd(&t->count[lens[s]], 1u);
    __syncthreads();
    // over-subscription, first codes, offsets (15 uniform steps)
    int left = 1, code = 0, off = 0;
    bool ok = true;
    for (int len = 1; len <= 15; ++len) {
        const int c = (int)uni(t->count[len]);
        left = (left << 1) - c;
        if (left < 0) ok = false;
        if (l == 0) {
            t->first[len] = (uint16_t)code;
            t->offs[len] = (uint16_t)off;
            t->longw[len] = ((uint32_t)(code + c) << 16) | ((uint32_t)(off - code) & 0xffffu);
        }
        code = (code + c) << 1;
        off += c;
    }
    if (!ok) return false;
    __syncthreads();
    // rank of every symbol within its length (symbol order), 64 symbols at a time
    uint32_t seen[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) seen[k] = 0u;
    for (int s0 = 0; s0 < nsym; s0 += 64) {
        const int s = s0 + l;
        const int len = s < nsym ? (int)lens[s] : 0;
        uint32_t rank = 0u;
#pragma unroll
        for (int k = 1; k <= 15; ++k) {
            const unsigned long long m = __ballot(len == k);
            if (len == k) rank = seen[k] + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
            seen[k] += (uint32_t)__popcll(m);
        }
        if (len > 0) {
            t->sym[t->offs[len] + rank] = (uint16_t)s;
            if (len <= TB) {
                const uint32_t c = (uint32_t)t->first[len] + rank;
                const uint32_t rev = __brev(c) >> (32 - len);
                const uint16_t e = (uint16_t)((s << 4) | len);
                for (uint32_t k = rev; k < (1u << TB); k += 1u << len) t->fast[k] = e;
            }
        }
    }
    __syncthreads();
    return true;
}

// one symbol (wave-uniform); -1: invalid code.  At least 15 valid bits in the buffer (zeros behind the stream's end).
template <int TB, int NS>
__device__ __forceinline__ int inf_decode(const InfTables<TB, NS> *t, InfBits &b) {
    const uint32_t e = uni(t->fast[b.peek(TB)]);
    if (e != 0u) {
        b.drop(e & 15u);
        return (int)(e >> 4);
    }
    // a code longer than the table (round 5: one word per length, all of them requested at once, instead of puff.c's
    // bit-by-bit walk with an LDS round trip per length -- 1 800 cycles for the 6 % of a record's literals whose codes
    // are longer than 10 bits, a third of a block's time).  A canonical code of `len` bits is the first len bits read
    // most significant first; codes of a length are consecutive from first[len] and greater than every shorter code
    // extended to that length.
    const uint32_t rv = __brev(b.peek(15)) >> 17;
    uint32_t w[15 - TB];
#pragma unroll
    for (int k = 0; k < 15 - TB; ++k) w[k] = t->longw[TB + 1 + k];
#pragma unroll
    for (int k = 0; k < 15 - TB; ++k) {
        const uint32_t len = (uint32_t)(TB + 1 + k), code = rv >> (15u - len), wk = uni(w[k]);
        if (code < (wk >> 16)) {
            b.drop(len);
            return (int)uni(t->sym[(int)code + (int)(int16_t)(wk & 0xffffu)]);
        }
    }
    return -1;
}

struct InfOut {
    uint8_t *dst;        // where stream r's bytes go (16-byte aligned)
    uint32_t cap;        // bytes kept
    uint32_t pos;        // bytes produced
    uint32_t flushed;    // bytes that have left the ring (multiple of INF_FLUSH)
    uint32_t a, b;       // Adler-32 of the flushed bytes
};
// bytes [flushed, flushed + m) of the ring -> global memory (those under cap), and into the check value
__device__ __forceinline__ void inf_flush(InfLds *L, InfOut &o, uint32_t m) {
    const int l = lane_id();
    const uint32_t j0 = (uint32_t)l * 16u;
    uint4 v = *reinterpret_cast<const uint4 *>(&L->win[(o.flushed + j0) & (INF_WIN - 1)]);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t s1 = 0u, s2 = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t j = j0 + (uint32_t)k;
        const uint32_t d = j < m ? ((w[k >> 2] >> (8 * (k & 3))) & 0xffu) : 0u;
        s1 += d;
        s2 += d * (m - j);
    }
    s1 = (uint32_t)wave_last_i(wave_incl_scan_i((int)s1));
    s2 = (uint32_t)wave_last_i(wave_incl_scan_i((int)s2));
    o.b = (o.b + m * o.a + s2) % 65521u;
    o.a = (o.a + s1) % 65521u;
    const uint32_t at = o.flushed + j0;
    if (at + 16u <= o.cap && j0 + 16u <= m) {
        *reinterpret_cast<uint4 *>(o.dst + at) = v;
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (at + (uint32_t)k < o.cap && j0 + (uint32_t)k < m) o.dst[at + k] = (uint8_t)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
        }
    }
    o.flushed += m;
}

// ---- a run of literals, decoded by the lanes together (round 5)
// A BLOW5 record is svb-zd bytes: 95 % of its DEFLATE symbols are literals, in runs of 18 (median 9) with codes of 6.4 bits
// on average, and the symbol loop above costs ~45 SCALAR instructions per symbol on the one scalar unit a compute unit's
// 19 streams share.  Here lane l looks up the code that would start at bit P + l of the stream (the 96 bits behind P are
// wave-uniform; one table read per lane), which gives every bit offset of the next 64 its successor; the offsets the
// stream really visits from P are found by pointer doubling across the lanes (symbol k's offset on lane k, four rounds
// of two ds_bpermute: up to 16 symbols), the leading run of literals among them is written to the ring by as many lanes
// at once, and the bit reader moves behind it.  A match, the end of the block, a code longer than the look-up table or
// the end of the 64-bit window ends the run; the symbol loop takes it from there.  ~45 scalar and ~60 vector / LDS
// instructions per run of typically 9 literals.  Returns the number of literals written.
constexpr int INF_RUN = 16;
#ifndef SGK_INF_RUN
#define SGK_INF_RUN 1   // 0: the symbol loop alone (development: A/B)
#endif
template <int TB, int NS>
__device__ __forceinline__ uint32_t inf_literal_run_at(InfLds *L, const InfTables<TB, NS> *t, const InfBits &b, InfOut &o,
                                                       uint32_t &P, bool &hit) {
    const int l = lane_id();
    const uint32_t k0 = P >> 5, sh = P & 31u;
    if ((k0 >> 6) != b.chunk) return ~0u;   // (the input window has just moved on: these few bits are the symbol loop's)
    const uint32_t w0 = b.peek_dword(k0), w1 = b.peek_dword(k0 + 1u), w2 = b.peek_dword(k0 + 2u), w3 = b.peek_dword(k0 + 3u);
    const uint32_t a0 = (uint32_t)((((unsigned long long)w1 << 32) | w0) >> sh);
    const uint32_t a1 = (uint32_t)((((unsigned long long)w2 << 32) | w1) >> sh);
    const uint32_t a2 = (uint32_t)((((unsigned long long)w3 << 32) | w2) >> sh);
    const uint32_t lo = l < 32 ? a0 : a1, hi = l < 32 ? a1 : a2;
    const uint32_t x = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)l & 31u);   // the bits from P + l on
    const uint32_t e = t->fast[x & ((1u << TB) - 1u)];                         // (symbol << 4) | length, 0: not in the table
    // successor of every offset (64: outside the window or unknown)
    int jp = e ? l + (int)(e & 15u) : 64;
    jp = jp > 64 ? 64 : jp;
    int pos = 0;   // lane k: the offset of the k-th symbol behind P
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = __builtin_amdgcn_ds_bpermute((pos & 63) << 2, jp);
        if ((l >> i) & 1) pos = pos >= 64 ? 64 : g;
        const int gg = __builtin_amdgcn_ds_bpermute((jp & 63) << 2, jp);
        jp = jp >= 64 ? 64 : gg;
    }
    const uint32_t ek = (uint32_t)__builtin_amdgcn_ds_bpermute((pos & 63) << 2, (int)e);
    const bool lit = l < INF_RUN && pos < 64 && ek != 0u && (ek >> 4) < 256u;
    const unsigned long long bad = __ballot(!lit);   // (never zero: lanes from INF_RUN on)
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(__ffsll((long long)bad) - 1);
    // hit: the run ended at a symbol that is no table literal (a match, the end of the block, a long code) -- the symbol
    // loop's next; otherwise at the end of the window or after INF_RUN literals, and another run may follow
    hit = n < (uint32_t)INF_RUN && __builtin_amdgcn_readlane(pos, (int)n) < 64;
    if (n == 0u) return 0u;
    if ((uint32_t)l < n) L->win[(o.pos + (uint32_t)l) & (INF_WIN - 1)] = (uint8_t)(ek >> 4);
    const uint32_t adv = (uint32_t)__builtin_amdgcn_readlane(pos + (int)(ek & 15u), (int)(n - 1u));
    o.pos += n;
    P += adv;
    return n;
}
// runs one behind the other from the bit reader's position, while they end at the window's end and the ring has room; the
// bit reader is moved once, behind the last.  Returns the literals written, ~0u: none, because the input window is moving.
template <int TB, int NS>
__device__ __forceinline__ uint32_t inf_literal_run(InfLds *L, const InfTables<TB, NS> *t, InfBits &b, InfOut &o, bool &hit) {
    uint32_t P = b.bit_pos(), total = 0u;
    for (;;) {
        const uint32_t n = inf_literal_run_at(L, t, b, o, P, hit);
        if (n == ~0u) {
            if (total == 0u) return ~0u;
            hit = false;
            break;
        }
        total += n;
        if (n == 0u || hit || o.pos - o.flushed >= (uint32_t)INF_FLUSH) break;
    }
    if (total) b.seek_bit(P);
    return total;
}

__global__ __launch_bounds__(64) void k_inflate(InfArgs a) {
    __shared__ __attribute__((aligned(16))) InfLds L;
    const uint32_t r = blockIdx.x;
    if (r >= a.n) return;
    const int l = lane_id();
    const uint8_t *src = a.in + a.in_offsets[r];
    const uint32_t in_len = a.in_lengths[r];
    const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 3u);
    InfBits b;
    b.base = reinterpret_cast<const uint32_t *>(src - lead);
    b.n_dw = (lead + in_len + 3u) / 4u;
    b.end_bit = (lead + in_len) * 8u;
    b.win = 0u;
    b.nxt = 0u;
    b.chunk = 0xfffffff0u;
    InfOut o;
    o.dst = a.out + a.out_offsets[r];
    o.cap = a.out_caps[r];
    o.pos = 0u;
    o.flushed = 0u;
    o.a = 1u;
    o.b = 0u;
    uint32_t st = INF_OK;
    if (in_len < 6u) st = INF_ERR_TRUNCATED;
    if (st == INF_OK) {
        b.seek_byte(lead);
        const uint32_t cmf = b.get(8), flg = b.get(8);
        if ((cmf & 15u) != 8u || (cmf >> 4) > 7u || ((cmf << 8) | flg) % 31u != 0u || (flg & 32u)) st = INF_ERR_HEADER;
    }
    bool last = false;
    while (st == INF_OK && !last) {
        b.refill();
        last = b.get(1) != 0u;
        const uint32_t type = b.get(2);
        if (type == 0u) {
            // stored: to the next byte, LEN, ~LEN, LEN bytes
            b.drop(b.cnt & 7u);
            b.refill();
            const uint32_t len = b.get(16);
            b.refill();
            const uint32_t nlen = b.get(16);
            if ((len ^ nlen) != 0xffffu) { st = INF_ERR_BLOCK; break; }
            const uint32_t byte0 = b.bit_pos() >> 3;   // (a whole byte: the buffer was byte aligned)
            if ((byte0 + len) * 8u > b.end_bit) { st = INF_ERR_TRUNCATED; break; }
            const uint8_t *p = reinterpret_cast<const uint8_t *>(b.base) + byte0;
            for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                const uint32_t i = i0 + (uint32_t)l;
                if (i < len) L.win[(o.pos + (uint32_t)l) & (INF_WIN - 1)] = p[i];
                __syncthreads();
                o.pos += len - i0 < 64u ? len - i0 : 64u;
                while (o.pos - o.flushed >= (uint32_t)INF_FLUSH) inf_flush(&L, o, INF_FLUSH);
            }
            if (o.pos > o.cap) { st = INF_ERR_ROOM; break; }
            b.seek_byte(byte0 + len);
            continue;
        }
        if (type == 3u) { st = INF_ERR_BLOCK; break; }
        int nlit, ndist;
        if (type == 1u) {
            nlit = 288;
            ndist = 30;
            for (int s = l; s < 288; s += 64) L.lens[s] = (uint8_t)(s < 144 ? 8 : (s < 256 ? 9 : (s < 280 ? 7 : 8)));
            if (l < 32) L.lens[288 + l] = (uint8_t)(l < 30 ? 5 : 0);
            __syncthreads();
        } else {
            b.refill();
            nlit = (int)b.get(5) + 257;
            ndist = (int)b.get(5) + 1;
            const int ncl = (int)b.get(4) + 4;
            if (nlit > 286 || ndist > 30) { st = INF_ERR_LENGTHS; break; }
            if (l < 19) L.lens[l] = 0;
            __syncthreads();
            for (int k = 0; k < ncl; ++k) {
                b.refill();
                const uint32_t v = b.get(3);
                if (l == 0) L.lens[INF_CLORDER[k]] = (uint8_t)v;
            }
            __syncthreads();
            if (!inf_build(&L.cl, L.lens, 19)) { st = INF_ERR_LENGTHS; break; }
            // the code lengths of the two alphabets, with repeats (serial)
            int i = 0, prev = 0;
            bool bad = false;
            // (lens[] is being rewritten while cl's table is in use: the cl lengths are no longer needed)
            __syncthreads();
            while (i < nlit + ndist) {
                b.refill();
                const int s = inf_decode(&L.cl, b);
                if (s < 0) { bad = true; break; }
                int rep = 1, val = s;
                if (s == 16) {
                    if (i == 0) { bad = true; break; }
                    val = prev;
                    rep = 3 + (int)b.get(2);
                } else if (s == 17) {
                    val = 0;
                    rep = 3 + (int)b.get(3);
                } else if (s == 18) {
                    val = 0;
                    rep = 11 + (int)b.get(7);
                }
                if (i + rep > nlit + ndist) { bad = true; break; }
                if (l < rep) L.lens[i + l] = (uint8_t)val;   // (rep <= 138: up to three lanes' worth)
                if (l + 64 < rep) L.lens[i + l + 64] = (uint8_t)val;
                if (l + 128 < rep) L.lens[i + l + 128] = (uint8_t)val;
                i += rep;
                prev = val;
            }
            __syncthreads();
            if (bad || b.overrun()) { st = bad ? INF_ERR_LENGTHS : INF_ERR_TRUNCATED; break; }
            if (uni(L.lens[256]) == 0u) { st = INF_ERR_LENGTHS; break; }
        }
        if (!inf_build(&L.lit, L.lens, nlit)) { st = INF_ERR_LENGTHS; break; }
        if (!inf_build(&L.dist, L.lens + nlit, ndist)) { st = INF_ERR_LENGTHS; break; }
        // ---- the symbols of the block
        uint32_t misses = 0u, pause = 0u;   // the literal runs pause where matches follow each other
        bool try_run = true;
#ifdef SGK_INF_STATS
        uint32_t st_runs = 0, st_lits = 0, st_zero = 0, st_moved = 0, st_serial = 0, st_paused = 0;
        unsigned long long st_trun = 0, st_tser = 0, st_t0 = 0, st_tblock = __builtin_amdgcn_s_memtime();
#define INF_T0() (st_t0 = __builtin_amdgcn_s_memtime())
#define INF_T1(acc) (acc += __builtin_amdgcn_s_memtime() - st_t0)
#define INF_STAT(x) (++x)
#define INF_STAT_ADD(x, v) (x += (v))
#else
#define INF_STAT(x)
#define INF_STAT_ADD(x, v)
#define INF_T0()
#define INF_T1(acc)
#endif
        for (;;) {
            b.refill();
            if (SGK_INF_RUN && pause == 0u && try_run) {
                bool hit = false;
                INF_T0();
                const uint32_t nrun = inf_literal_run(&L, &L.lit, b, o, hit);
                INF_T1(st_trun);
                if (nrun == ~0u) {
                    INF_STAT(st_moved);
                } else if (nrun) {
                    INF_STAT(st_runs);
                    INF_STAT_ADD(st_lits, nrun);
                    misses = 0u;
                    if (o.pos - o.flushed >= (uint32_t)INF_FLUSH) {
                        if (o.pos > o.cap) { st = INF_ERR_ROOM; break; }
                        if (b.overrun()) { st = INF_ERR_TRUNCATED; break; }
                        __syncthreads();
                        while (o.pos - o.flushed >= (uint32_t)INF_FLUSH) inf_flush(&L, o, INF_FLUSH);
                    }
                    if (!hit) continue;   // (else the symbol loop takes the symbol that ended the run, without another try)
                }
                else {
                    INF_STAT(st_zero);
                    if (++misses >= 4u) { pause = 32u; misses = 0u; }
                }
            } else if (pause) { --pause; INF_STAT(st_paused); }
            try_run = true;
            INF_STAT(st_serial);
            INF_T0();
            const int s = inf_decode(&L.lit, b);
            if (s < 256) {
                if (s < 0) { st = INF_ERR_CODE; break; }
                if (l == 0) L.win[o.pos & (INF_WIN - 1)] = (uint8_t)s;
                ++o.pos;
            } else {
                if (s == 256) break;
                if (s > 285) { st = INF_ERR_CODE; break; }
                // length and distance bases / extra bits by arithmetic (RFC 1951 3.2.5), not from a table: the table
                // look-ups were global loads on the symbol loop's critical path
                uint32_t len;
                if (s < 265) len = (uint32_t)s - 254u;
                else if (s == 285) len = 258u;
                else {
                    const uint32_t q = (uint32_t)s - 261u, e = q >> 2;
                    len = ((4u + (q & 3u)) << e) + 3u + b.get(e);
                }
                b.refill();
                const int ds = inf_decode(&L.dist, b);
                if (ds < 0 || ds > 29) { st = INF_ERR_CODE; break; }
                b.refill();
                uint32_t dist;
                if (ds < 4) dist = (uint32_t)ds + 1u;
                else {
                    const uint32_t e = ((uint32_t)ds >> 1) - 1u;
                    dist = ((2u + ((uint32_t)ds & 1u)) << e) + 1u + b.get(e);
                }
                if (dist > o.pos) { st = INF_ERR_DISTANCE; break; }
                // out[pos + i] = out[pos - dist + (i mod dist)]: every source byte lies in front of pos
                __syncthreads();   // (the literals lane 0 wrote are in the ring before other lanes read them)
                if (dist <= (uint32_t)INF_NEAR) {
                    if (dist >= len) {   // (wave-uniform: the usual match, no lane needs i mod dist -- an integer division)
                        for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                            const uint32_t i = i0 + (uint32_t)l;
                            if (i < len) L.win[(o.pos + i) & (INF_WIN - 1)] = L.win[(o.pos - dist + i) & (INF_WIN - 1)];
                        }
                    } else {
                        for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                            const uint32_t i = i0 + (uint32_t)l;
                            if (i < len) L.win[(o.pos + i) & (INF_WIN - 1)] = L.win[(o.pos - dist + i % dist) & (INF_WIN - 1)];
                        }
                    }
                } else {
                    // further back than the ring holds: those bytes left it (inf_flush) at least INF_NEAR - 1024 - 258
                    // bytes ago.  Read past the vector L1 (the wave's own stores went through it to L2), behind them.
                    if (o.pos - dist + len > o.cap) { st = INF_ERR_ROOM; break; }   // (never: pos <= cap is checked below)
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    for (uint32_t i0 = 0; i0 < len; i0 += 64u) {
                        const uint32_t i = i0 + (uint32_t)l;
                        if (i < len) {
                            const uint8_t v = __hip_atomic_load(o.dst + (o.pos - dist + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (dist > len here)
                            L.win[(o.pos + i) & (INF_WIN - 1)] = v;
                        }
                    }
                }
                __syncthreads();
                o.pos += len;
            }
            INF_T1(st_tser);
            if (o.pos - o.flushed >= (uint32_t)INF_FLUSH) {
                // (once per KB of output: the room and the input's end are looked at here, not per symbol.  Behind the
                // input's end the reader yields zero bits: at most a KB of output is decoded from them before this
                // notices; a far match never reads beyond what has been flushed, and that is under `cap` here.)
                if (o.pos > o.cap) { st = INF_ERR_ROOM; break; }
                if (b.overrun()) { st = INF_ERR_TRUNCATED; break; }
                __syncthreads();
                while (o.pos - o.flushed >= (uint32_t)INF_FLUSH) inf_flush(&L, o, INF_FLUSH);
            }
        }
#ifdef SGK_INF_STATS
        if (r == 0 && l == 0)
            printf("block: runs %u literals-in-runs %u zero-runs %u window-moved %u serial %u (paused %u); 100 MHz ticks: runs %llu serial %llu block %llu\n", st_runs, st_lits,
                   st_zero, st_moved, st_serial, st_paused, st_trun, st_tser, (unsigned long long)__builtin_amdgcn_s_memtime() - st_tblock);
#endif
        if (st == INF_OK && b.overrun()) st = INF_ERR_TRUNCATED;
        if (st == INF_OK && o.pos > o.cap) st = INF_ERR_ROOM;
    }
    if (st == INF_OK) {
        __syncthreads();
        while (o.pos - o.flushed >= (uint32_t)INF_FLUSH) inf_flush(&L, o, INF_FLUSH);
        if (o.pos > o.flushed) inf_flush(&L, o, o.pos - o.flushed);
        // the check value: four bytes, most significant first, on the next byte boundary
        b.drop(b.cnt & 7u);
        b.refill();
        uint32_t want = 0u;
        for (int k = 0; k < 4; ++k) {
            b.refill();
            want = (want << 8) | b.get(8);
        }
        if (b.overrun()) st = INF_ERR_TRUNCATED;
        else if (want != ((o.b << 16) | o.a)) st = INF_ERR_ADLER;
    }
    if (l == 0) {
        a.status[r] = st;
        a.out_lengths[r] = o.pos;
    }
}

int launch_inflate(const InfArgs &a, hipStream_t st) {
    if (a.n == 0) return SGK_OK;
    ProfScope ps("k_inflate", st);
    hipLaunchKernelGGL(k_inflate, dim3(a.n), dim3(64), 0, st, a);
    SGK_HIP_TRY(hipGetLastError());
    return SGK_OK;
}

}  // namespace sgk

extern "C" int sgk_inflate(const uint8_t *in, const uint64_t *in_offsets, const uint32_t *in_lengths, uint32_t n, uint8_t *out,
                           const uint64_t *out_offsets, const uint32_t *out_caps, uint32_t *out_lengths, uint32_t *status,
                           void *stream) {
    if (n == 0) return SGK_OK;
    if (!in || !in_offsets || !in_lengths || !out || !out_offsets || !out_caps || !out_lengths || !status) return SGK_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(out) & 15u) return SGK_ERR_ALIGN;
    sgk::InfArgs a;
    a.in = in; a.in_offsets = in_offsets; a.in_lengths = in_lengths; a.out = out; a.out_offsets = out_offsets;
    a.out_caps = out_caps; a.out_lengths = out_lengths; a.status = status; a.n = n;
    return sgk::launch_inflate(a, static_cast<hipStream_t>(stream));
}
