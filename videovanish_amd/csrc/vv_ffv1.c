/* vv_ffv1.c -- host-side FFV1 (RFC 9043) codec for the frame I/O on either side of the hot path: version 3 encoder (intra), version 0 / 1 / 3
 * decoder incl. non-key frames
 * (reference tools.py:30-45 writes FFV1 in Matroska through cv2/ffmpeg).  Plain C, no GPU: codec I/O is CPU
 * work outside the timed path.  Built into libvvio.so by build.sh; bound by videovanish_amd/frameio.py.
 *
 * What is implemented (the subset ffmpeg's encoder emits for 8-bit packed RGB, and what this encoder writes):
 *   version 3, micro_version 4; coder_type 0 (Golomb-Rice sample coding, range coder for the headers); colorspace_type 1
 *   (RGB through the reversible JPEG 2000 RCT, planes G, B-G, R-G coded with 9 bits); no alpha; num_h_slices = 1,
 *   num_v_slices >= 1 (horizontal bands); ec = 1 (CRC-32 per slice and on the configuration record); intra = 1.
 * The decoder additionally accepts: any quantisation-table set with 3 or 5 context inputs (states_coded = 0); num_h_slices > 1
 * (the 2 x 2 slice grid libavcodec picks by default); range-coded sample data -- coder_type 1 (RFC 9043 3.8.1.5
 * default_state_transition = the table build_rac_states() computes, checked against the RFC's listing in the tests) and
 * coder_type 2 (custom table: deltas to the default in the configuration record); an extra (alpha) plane, decoded and dropped.
 * Round 4: planar YCbCr streams (round 3), FFV1 VERSION 0 / 1 streams (no configuration record: the parameters and ONE quantisation-table
 * set travel in the range-coded header of every key frame, one slice per frame, no slice header / footer / CRC -- what libavcodec's encoder
 * picks by itself for frames up to 720 x 576) and NON-KEY frames of every version (a frame whose key-frame bit is 0 carries no header and
 * continues from the adaptive context states the previous frame left behind, slice by slice -- what an encoder with gop_size > 1 emits,
 * e.g. cv2.VideoWriter's default of 12 [UNVERIFIED-3P]) through the stateful decoder object (vvio_ffv1_decoder_*).
 * Refused with an error code: more than 8 bits per sample, coded initial states, version 2 (experimental, never released) and versions
 * above 3 (different slice header / context handling), a non-key frame without a preceding key frame.  Every header field read from the
 * file is range-checked before it is used.
 *
 * PARITY UNPINNED against a real FFV1 decoder (no ffmpeg / cv2 in the build image): restated from RFC 9043 and the public
 * libavcodec ffv1 sources; pinned here only by lossless round trips and structural checks (tests/test_frameio_cpu.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CONTEXT_SIZE 32
#define MAX_CTX 32768

/* ---------------------------------------------------------------------------------------------------------------- CRC */
/* CRC-32, polynomial 0x04C11DB7, MSB first, init 0, no final xor; stored big-endian so that crc(data || crc) == 0 */
static uint32_t crc_table[256];
static int crc_ready;
static void crc_init(void) {
    for (int i = 0; i < 256; ++i) {
        uint32_t c = (uint32_t)i << 24;
        for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
        crc_table[i] = c;
    }
    crc_ready = 1;
}
static uint32_t crc32_mpeg(const uint8_t* p, int n) {
    if (!crc_ready) crc_init();
    uint32_t c = 0;
    for (int i = 0; i < n; ++i) c = (c << 8) ^ crc_table[(c >> 24) ^ p[i]];
    return c;
}
static void put_be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

/* -------------------------------------------------------------------------------------------------------- range coder */
typedef struct {
    int low, range, outstanding_count, outstanding_byte;
    uint8_t zero_state[256], one_state[256];
    uint8_t *start, *ptr, *end;
    int overflow;
} RangeCoder;

/* state transition table of the HEADER coder (libavcodec ff_build_rac_states(c, 0.05 * 2^32, 256 - 8)) */
static void build_rac_states(RangeCoder* c) {
    const int64_t one = 1LL << 32;
    const int64_t factor = (int64_t)(0.05 * (double)(1LL << 32));
    const int max_p = 256 - 8;
    int64_t p;
    int last_p8 = 0, p8, i;
    memset(c->zero_state, 0, 256);
    memset(c->one_state, 0, 256);
    p = one / 2;
    for (i = 0; i < 128; i++) {
        p8 = (int)((256 * p + one / 2) >> 32);
        if (p8 <= last_p8) p8 = last_p8 + 1;
        if (last_p8 && last_p8 < 256 && p8 <= max_p) c->one_state[last_p8] = (uint8_t)p8;
        p += ((one - p) * factor + one / 2) >> 32;
        last_p8 = p8;
    }
    for (i = 256 - max_p; i <= max_p; i++) {
        if (c->one_state[i]) continue;
        p = (i * one + 128) >> 8;
        p += ((one - p) * factor + one / 2) >> 32;
        p8 = (int)((256 * p + one / 2) >> 32);
        if (p8 <= i) p8 = i + 1;
        if (p8 > max_p) p8 = max_p;
        c->one_state[i] = (uint8_t)p8;
    }
    for (i = 1; i < 255; i++) c->zero_state[i] = (uint8_t)(256 - c->one_state[256 - i]);
}

static void rc_init_enc(RangeCoder* c, uint8_t* buf, int size) {
    c->start = c->ptr = buf; c->end = buf + size;
    c->low = 0; c->range = 0xFF00; c->outstanding_count = 0; c->outstanding_byte = -1; c->overflow = 0;
    build_rac_states(c);
}
static void rc_out(RangeCoder* c, int b) { if (c->ptr < c->end) *c->ptr++ = (uint8_t)b; else c->overflow = 1; }
static void rc_renorm_enc(RangeCoder* c) {
    while (c->range < 0x100) {
        if (c->outstanding_byte < 0) {
            c->outstanding_byte = c->low >> 8;
        } else if (c->low <= 0xFF00) {
            rc_out(c, c->outstanding_byte);
            for (; c->outstanding_count; c->outstanding_count--) rc_out(c, 0xFF);
            c->outstanding_byte = c->low >> 8;
        } else if (c->low >= 0x10000) {
            rc_out(c, c->outstanding_byte + 1);
            for (; c->outstanding_count; c->outstanding_count--) rc_out(c, 0x00);
            c->outstanding_byte = (c->low >> 8) & 0xFF;
        } else {
            c->outstanding_count++;
        }
        c->low = (c->low & 0xFF) << 8;
        c->range <<= 8;
    }
}
static void put_rac(RangeCoder* c, uint8_t* state, int bit) {
    int range1 = (c->range * (*state)) >> 8;
    if (!bit) { c->range -= range1; *state = c->zero_state[*state]; }
    else { c->low += c->range - range1; c->range = range1; *state = c->one_state[*state]; }
    rc_renorm_enc(c);
}
/* ff_rac_terminate(c, version): version 1 first codes a 0 with state 129 (the Golomb-Rice hand-over bit of FFV1 >= 3.2) */
static int rc_terminate(RangeCoder* c, int version) {
    if (version == 1) { uint8_t st = 129; put_rac(c, &st, 0); }
    c->range = 0xFF; c->low += 0xFF; rc_renorm_enc(c);
    c->range = 0xFF; rc_renorm_enc(c);
    return (int)(c->ptr - c->start);
}
static void put_symbol(RangeCoder* c, uint8_t* state, int v, int is_signed) {
    if (v) {
        const int a = v < 0 ? -v : v;
        int e = 0, i;
        while ((a >> (e + 1)) != 0) e++;          /* floor(log2(a)) */
        put_rac(c, state + 0, 0);
        for (i = 0; i < e; i++) put_rac(c, state + 1 + (i < 9 ? i : 9), 1);
        put_rac(c, state + 1 + (e < 9 ? e : 9), 0);
        for (i = e - 1; i >= 0; i--) put_rac(c, state + 22 + (i < 9 ? i : 9), (a >> i) & 1);
        if (is_signed) put_rac(c, state + 11 + (e < 10 ? e : 10), v < 0);
    } else {
        put_rac(c, state + 0, 1);
    }
}

static void rc_init_dec(RangeCoder* c, const uint8_t* buf, int size) {
    c->start = (uint8_t*)buf; c->ptr = (uint8_t*)buf; c->end = (uint8_t*)buf + size;
    c->overflow = 0;
    build_rac_states(c);
    c->low = size >= 2 ? ((buf[0] << 8) | buf[1]) : 0;
    c->ptr += 2;
    c->range = 0xFF00;
    if (c->low >= 0xFF00) { c->low = 0xFF00; c->end = c->ptr; }
}
static void rc_refill(RangeCoder* c) {
    if (c->range < 0x100) {
        c->range <<= 8; c->low <<= 8;
        if (c->ptr < c->end) { c->low += *c->ptr; c->ptr++; }
        else c->overflow++;
    }
}
static int get_rac(RangeCoder* c, uint8_t* state) {
    int range1 = (c->range * (*state)) >> 8;
    c->range -= range1;
    if (c->low < c->range) { *state = c->zero_state[*state]; rc_refill(c); return 0; }
    c->low -= c->range; *state = c->one_state[*state]; c->range = range1; rc_refill(c); return 1;
}
static int get_symbol(RangeCoder* c, uint8_t* state, int is_signed, int* err) {
    if (get_rac(c, state + 0)) return 0;
    int e = 0, i;
    unsigned a = 1;
    while (get_rac(c, state + 1 + (e < 9 ? e : 9))) { if (++e > 31) { *err = 1; return 0; } }
    for (i = e - 1; i >= 0; i--) a += a + get_rac(c, state + 22 + (i < 9 ? i : 9));
    int s = -(is_signed && get_rac(c, state + 11 + (e < 10 ? e : 10)));
    if (a > 0x7FFFFFFFu) { *err = 1; return 0; }       /* e == 31: does not fit an int (a crafted header) */
    return (int)((a ^ (unsigned)s) - (unsigned)s);
}

/* ---------------------------------------------------------------------------------------------- Golomb-Rice bit I/O */
typedef struct { uint8_t* buf; int cap; int64_t bitpos; int overflow; } BitW;
static void bw_put(BitW* w, int n, uint32_t v) {          /* n <= 32 bits, MSB first */
    for (int i = n - 1; i >= 0; --i) {
        const int64_t byte = w->bitpos >> 3;
        if (byte >= w->cap) { w->overflow = 1; w->bitpos++; continue; }
        if ((w->bitpos & 7) == 0) w->buf[byte] = 0;
        w->buf[byte] |= (uint8_t)(((v >> i) & 1u) << (7 - (w->bitpos & 7)));
        w->bitpos++;
    }
}
typedef struct { const uint8_t* buf; int64_t nbits; int64_t pos; int overflow; } BitR;
static int br_get1(BitR* r) {
    if (r->pos >= r->nbits) { r->overflow = 1; r->pos++; return 0; }
    const int b = (r->buf[r->pos >> 3] >> (7 - (r->pos & 7))) & 1;
    r->pos++;
    return b;
}
static uint32_t br_get(BitR* r, int n) { uint32_t v = 0; for (int i = 0; i < n; ++i) v = (v << 1) | (uint32_t)br_get1(r); return v; }

typedef struct { int16_t drift; uint16_t error_sum; int8_t bias; uint8_t count; } VlcState;
static const uint8_t log2_run[41] = {0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7,
                                     8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24};

static int fold(int diff, int bits) {           /* sign-extend the low `bits` bits */
    const int sh = 32 - bits;
    return (int)((uint32_t)diff << sh) >> sh;
}
static void update_vlc_state(VlcState* st, int v) {
    int drift = st->drift, count = st->count;
    st->error_sum += (uint16_t)(v < 0 ? -v : v);
    drift += v;
    if (count == 128) { count >>= 1; drift >>= 1; st->error_sum >>= 1; }
    count++;
    if (drift <= -count) {
        st->bias = (int8_t)(st->bias - 1 < -128 ? -128 : st->bias - 1);
        drift = drift + count > -count + 1 ? drift + count : -count + 1;
    } else if (drift > 0) {
        st->bias = (int8_t)(st->bias + 1 > 127 ? 127 : st->bias + 1);
        drift = drift - count < 0 ? drift - count : 0;
    }
    st->drift = (int16_t)drift; st->count = (uint8_t)count;
}
static int vlc_k(const VlcState* st) { int i = st->count, k = 0; while (i < st->error_sum) { k++; i += i; } return k; }
static void put_vlc_symbol(BitW* w, VlcState* st, int v, int bits) {
    v = fold(v - st->bias, bits);
    const int k = vlc_k(st);
    const int code = v ^ ((2 * st->drift + st->count) >> 31);
    /* set_sr_golomb(code, k, limit 12, esc_len bits) */
    int u = -2 * code - 1; u ^= (u >> 31);
    const int e = u >> k;
    if (e < 12) bw_put(w, e + k + 1, (1u << k) + ((unsigned)u & ((1u << k) - 1u)));
    else { bw_put(w, 12, 0); bw_put(w, bits, (uint32_t)(u - 12 + 1)); }
    update_vlc_state(st, v);
}
static int get_vlc_symbol(BitR* r, VlcState* st, int bits) {
    const int k = vlc_k(st);
    int e = 0, u;
    while (e < 12 && br_get1(r) == 0) e++;
    if (e < 12) u = (e << k) + (int)br_get(r, k);
    else u = (int)br_get(r, bits) + 12 - 1;
    int v = (u >> 1) ^ -(u & 1);
    v ^= ((2 * st->drift + st->count) >> 31);
    const int ret = fold(v + st->bias, bits);
    update_vlc_state(st, v);
    return ret;
}

/* ------------------------------------------------------------------------------------------------------ codec state */
typedef struct {
    int16_t quant[5][256];
    int context_count;
    int five;                         /* quant[3] / quant[4] in use */
} QuantSet;

static inline int mid_pred(int a, int b, int c) {
    if (a > b) { if (c > b) { if (c > a) b = a; else b = c; } }
    else { if (b > c) { if (c > a) b = c; else b = a; } }
    return b;
}
static inline int get_context(const QuantSet* q, const int16_t* src, const int16_t* last, const int16_t* last2) {
    const int LT = last[-1], T = last[0], RT = last[1], L = src[-1];
    int c = q->quant[0][(L - LT) & 0xFF] + q->quant[1][(LT - T) & 0xFF] + q->quant[2][(T - RT) & 0xFF];
    if (q->five) c += q->quant[3][(src[-2] - L) & 0xFF] + q->quant[4][(last2[0] - T) & 0xFF];
    return c;
}

/* the encoder's own table set: 9 levels per input (0 | 1..2 | 3..6 | 7..14 | 15..), three inputs -> (9^3 + 1) / 2 = 365 contexts */
static void default_quant(QuantSet* q) {
    static const int edge[4] = {1, 3, 7, 15};
    int scale = 1;
    memset(q, 0, sizeof(*q));
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < 128; ++i) { int v = 0; for (int e = 0; e < 4; ++e) if (i >= edge[e]) v = e + 1; q->quant[t][i] = (int16_t)(scale * v); }
        for (int i = 1; i < 128; ++i) q->quant[t][256 - i] = (int16_t)-q->quant[t][i];
        q->quant[t][128] = (int16_t)-q->quant[t][127];
        scale *= 9;
    }
    q->context_count = (scale + 1) / 2;
    q->five = 0;
}
static void write_quant_table(RangeCoder* c, const int16_t* tab) {
    uint8_t state[CONTEXT_SIZE];
    int last = 0, i;
    memset(state, 128, sizeof(state));
    for (i = 1; i < 128; i++)
        if (tab[i] != tab[i - 1]) { put_symbol(c, state, i - last - 1, 0); last = i; }
    put_symbol(c, state, i - last - 1, 0);
}
static int read_quant_table(RangeCoder* c, int16_t* tab, int scale) {
    uint8_t state[CONTEXT_SIZE];
    int v, i = 0, err = 0;
    memset(state, 128, sizeof(state));
    for (v = 0; i < 128; v++) {
        unsigned len = (unsigned)get_symbol(c, state, 0, &err) + 1u;
        if (err || len > (unsigned)(128 - i) || !len) return -1;
        while (len--) { tab[i] = (int16_t)(scale * v); i++; }
    }
    for (i = 1; i < 128; i++) tab[256 - i] = (int16_t)-tab[i];
    tab[128] = (int16_t)-tab[127];
    return 2 * v - 1;
}

/* ----------------------------------------------------------------------------------------------- configuration record */
typedef struct {
    int version, micro, coder, colorspace, bits, chroma_planes, hshift, vshift, alpha, nh, nv, nsets, ec, intra;
    uint8_t one_state[256];           /* coder_type 2: the custom state transition table (RFC 9043 4.2.3 state_transition_delta) */
    QuantSet sets[8];
} Config;

int vvio_ffv1_config_record(int num_v_slices, uint8_t* out, int cap) {
    RangeCoder c;
    uint8_t state[CONTEXT_SIZE];
    QuantSet q;
    if (cap < 64 || num_v_slices < 1) return -1;
    default_quant(&q);
    memset(state, 128, sizeof(state));
    rc_init_enc(&c, out, cap - 4);
    put_symbol(&c, state, 3, 0);               /* version */
    put_symbol(&c, state, 4, 0);               /* micro_version */
    put_symbol(&c, state, 0, 0);               /* coder_type: Golomb-Rice */
    put_symbol(&c, state, 1, 0);               /* colorspace_type: RGB (JPEG 2000 RCT) */
    put_symbol(&c, state, 8, 0);               /* bits_per_raw_sample */
    put_rac(&c, state, 1);                     /* chroma_planes */
    put_symbol(&c, state, 0, 0);               /* log2_h_chroma_subsample */
    put_symbol(&c, state, 0, 0);               /* log2_v_chroma_subsample */
    put_rac(&c, state, 0);                     /* extra_plane (alpha) */
    put_symbol(&c, state, 0, 0);               /* num_h_slices - 1 */
    put_symbol(&c, state, num_v_slices - 1, 0);
    put_symbol(&c, state, 1, 0);               /* quant_table_set_count */
    for (int t = 0; t < 5; ++t) write_quant_table(&c, q.quant[t]);
    put_rac(&c, state, 0);                     /* states_coded = 0 for the set */
    put_symbol(&c, state, 1, 0);               /* ec */
    put_symbol(&c, state, 1, 0);               /* intra */
    const int n = rc_terminate(&c, 0);
    if (c.overflow) return -1;
    put_be32(out + n, crc32_mpeg(out, n));
    return n + 4;
}

static int parse_config(const uint8_t* rec, int len, Config* cf) {
    RangeCoder c;
    uint8_t state[CONTEXT_SIZE];
    int err = 0;
    if (len < 6) return -1;
    memset(cf, 0, sizeof(*cf));
    memset(state, 128, sizeof(state));
    rc_init_dec(&c, rec, len);
    cf->version = get_symbol(&c, state, 0, &err);
    if (cf->version < 3) return -2;            /* versions 0 / 1 have no configuration record (their header is in the key frames); 2 was never released */
    if (cf->version > 3) return -26;           /* version 4+: other slice header fields / context handling -- not decoded with version 3 rules */
    cf->micro = get_symbol(&c, state, 0, &err);
    if (err || cf->micro < 0 || cf->micro > 4) return -26;     /* micro_version beyond what RFC 9043 describes */
    cf->coder = get_symbol(&c, state, 0, &err);
    if (err || cf->coder < 0 || cf->coder > 2) return -3;
    if (cf->coder == 2) {                      /* custom table = default_state_transition (what build_rac_states made) + coded deltas */
        for (int i = 1; i < 256; ++i) {
            const int v = get_symbol(&c, state, 1, &err) + c.one_state[i];
            if (err || v < 1 || v > 255) return -3;
            cf->one_state[i] = (uint8_t)v;
        }
    }
    cf->colorspace = get_symbol(&c, state, 0, &err);
    cf->bits = get_symbol(&c, state, 0, &err);
    cf->chroma_planes = get_rac(&c, state);
    cf->hshift = get_symbol(&c, state, 0, &err);
    cf->vshift = get_symbol(&c, state, 0, &err);
    cf->alpha = get_rac(&c, state);
    cf->nh = get_symbol(&c, state, 0, &err) + 1;
    cf->nv = get_symbol(&c, state, 0, &err) + 1;
    cf->nsets = get_symbol(&c, state, 0, &err);
    if (err || cf->nsets < 1 || cf->nsets > 8) return -4;
    for (int s = 0; s < cf->nsets; ++s) {
        int count = 1;
        for (int t = 0; t < 5; ++t) {
            const int r = read_quant_table(&c, cf->sets[s].quant[t], count);
            if (r < 0) return -5;
            count *= r;
            if (count > MAX_CTX) return -5;
        }
        cf->sets[s].context_count = (count + 1) / 2;
        cf->sets[s].five = cf->sets[s].quant[3][127] || cf->sets[s].quant[4][127];
    }
    for (int s = 0; s < cf->nsets; ++s)
        if (get_rac(&c, state)) return -6;     /* coded initial states: not supported */
    cf->ec = get_symbol(&c, state, 0, &err);
    if (cf->version > 2 && cf->micro > 2) cf->intra = get_symbol(&c, state, 0, &err);
    if (err) return -7;
    if ((cf->colorspace != 0 && cf->colorspace != 1) || (cf->bits != 0 && cf->bits != 8)) return -8;       /* 8 bits per sample only */
    if (cf->colorspace == 0 && (cf->hshift < 0 || cf->hshift > 2 || cf->vshift < 0 || cf->vshift > 2 || (cf->alpha && !cf->chroma_planes))) return -8;
    if (cf->nh < 1 || cf->nv < 1 || cf->nh > 64 || cf->nv > 64 || cf->ec < 0 || cf->ec > 2) return -8;
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ slices */
static void slice_rows(int H, int nv, int sy, int* y0, int* h) {
    const int a = (int)((int64_t)sy * H / nv), b = (int)((int64_t)(sy + 1) * H / nv);
    *y0 = a; *h = b - a;
}

/* one slice (rows y0 .. y0+h-1 of an RGB24 image) -> bytes written, or -1 */
static int encode_slice(const uint8_t* rgb, int W, int stride, int y0, int h, int sy, int first, const QuantSet* q, uint8_t* out, int cap) {
    RangeCoder c;
    uint8_t state[CONTEXT_SIZE];
    rc_init_enc(&c, out, cap);
    if (first) { uint8_t keystate = 128; put_rac(&c, &keystate, 1); }     /* key frame flag lives in slice 0's coder */
    memset(state, 128, sizeof(state));
    put_symbol(&c, state, 0, 0);               /* slice_x */
    put_symbol(&c, state, sy, 0);              /* slice_y */
    put_symbol(&c, state, 0, 0);               /* slice_width - 1 (in slice units) */
    put_symbol(&c, state, 0, 0);               /* slice_height - 1 */
    put_symbol(&c, state, 0, 0);               /* quant_table_set_index, plane 0 */
    put_symbol(&c, state, 0, 0);               /* ... plane 1 (chroma) */
    put_symbol(&c, state, 3, 0);               /* picture_structure: progressive */
    put_symbol(&c, state, 0, 0);               /* sar_num */
    put_symbol(&c, state, 0, 0);               /* sar_den */
    const int ac_bytes = rc_terminate(&c, 1);
    if (c.overflow) return -1;
    BitW bw = {out + ac_bytes, cap - ac_bytes, 0, 0};

    VlcState* vlc[2];
    for (int p = 0; p < 2; ++p) {
        vlc[p] = (VlcState*)malloc(sizeof(VlcState) * (size_t)q->context_count);
        if (!vlc[p]) return -1;
        for (int i = 0; i < q->context_count; ++i) { vlc[p][i].drift = 0; vlc[p][i].error_sum = 4; vlc[p][i].bias = 0; vlc[p][i].count = 1; }
    }
    /* three planes (G, B-G, R-G), three lines each, 3 samples of margin on both sides */
    const int LW = W + 6;
    int16_t* lines = (int16_t*)calloc((size_t)3 * 3 * LW, sizeof(int16_t));
    if (!lines) { free(vlc[0]); free(vlc[1]); return -1; }
    int run_index = 0;
    for (int y = 0; y < h; ++y) {
        int16_t* s[3][3];
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 3; ++i) s[p][i] = lines + ((size_t)p * 3 + (size_t)((y + 3 - i) % 3)) * LW + 3;   /* [0] current, [1] above, [2] two above */
        const uint8_t* row = rgb + (size_t)(y0 + y) * stride;
        for (int x = 0; x < W; ++x) {
            int r = row[3 * x], g = row[3 * x + 1], b = row[3 * x + 2];
            b -= g; r -= g; g += (b + r) >> 2; b += 256; r += 256;
            s[0][0][x] = (int16_t)g; s[1][0][x] = (int16_t)b; s[2][0][x] = (int16_t)r;
        }
        for (int p = 0; p < 3; ++p) {
            int16_t *cur = s[p][0], *last = s[p][1], *last2 = s[p][2];
            VlcState* vs = vlc[(p + 1) / 2];
            if (y == 0) { memset(last - 3, 0, sizeof(int16_t) * LW); memset(last2 - 3, 0, sizeof(int16_t) * LW); }
            else if (y == 1) memset(last2 - 3, 0, sizeof(int16_t) * LW);
            cur[-1] = last[0]; last[W] = last[W - 1];
            cur[-2] = 0;             /* RFC 9043 3.2: the additional column to the left is 0 (only read by 5-input sets, never by this encoder's own tables) */
            int run_count = 0, run_mode = 0;
            for (int x = 0; x < W; ++x) {
                int context = get_context(q, cur + x, last + x, last2 + x);
                int diff = cur[x] - mid_pred(cur[x - 1], cur[x - 1] + last[x] - last[x - 1], last[x]);
                if (context < 0) { context = -context; diff = -diff; }
                diff = fold(diff, 9);
                if (context == 0) run_mode = 1;
                if (run_mode) {
                    if (diff) {
                        while (run_count >= 1 << log2_run[run_index]) { run_count -= 1 << log2_run[run_index]; run_index++; bw_put(&bw, 1, 1); }
                        bw_put(&bw, 1 + log2_run[run_index], (uint32_t)run_count);
                        if (run_index) run_index--;
                        run_count = 0; run_mode = 0;
                        if (diff > 0) diff--;
                    } else {
                        run_count++;
                    }
                }
                if (run_mode == 0) put_vlc_symbol(&bw, &vs[context], diff, 9);
            }
            if (run_mode) {
                while (run_count >= 1 << log2_run[run_index]) { run_count -= 1 << log2_run[run_index]; run_index++; bw_put(&bw, 1, 1); }
                if (run_count) bw_put(&bw, 1, 1);
            }
        }
    }
    free(lines); free(vlc[0]); free(vlc[1]);
    if (bw.overflow) return -1;
    int bytes = ac_bytes + (int)((bw.bitpos + 7) >> 3);
    if (bytes + 8 > cap || bytes >= (1 << 24)) return -1;
    out[bytes] = (uint8_t)(bytes >> 16); out[bytes + 1] = (uint8_t)(bytes >> 8); out[bytes + 2] = (uint8_t)bytes;   /* slice_size */
    bytes += 3;
    out[bytes++] = 0;                                                                                                /* error_status */
    put_be32(out + bytes, crc32_mpeg(out, bytes));
    return bytes + 4;
}

/* RGB24 frame -> FFV1 packet.  Returns the packet size, or -1 (buffer too small: cap >= 2 * W * H * 3 + 4096 is always enough) */
int vvio_ffv1_encode_frame(const uint8_t* rgb, int W, int H, int num_v_slices, uint8_t* out, int cap) {
    QuantSet q;
    int pos = 0;
    if (!rgb || !out || W < 1 || H < 1 || num_v_slices < 1 || num_v_slices > H) return -1;
    default_quant(&q);
    for (int sy = 0; sy < num_v_slices; ++sy) {
        int y0, h;
        slice_rows(H, num_v_slices, sy, &y0, &h);
        const int n = encode_slice(rgb, W, W * 3, y0, h, sy, sy == 0, &q, out + pos, cap - pos);
        if (n < 0) return -1;
        pos += n;
    }
    return pos;
}

/* one plane of a planar 8-bit (YCbCr) slice: RFC 9043 3.1-3.8 on unsigned 8-bit samples; the run index restarts with every plane, the
 * context states of a plane index are shared by the planes that use it (Cb and Cr) */
typedef struct { int ac; RangeCoder* c; BitR* br; } SampleSrc;
static int decode_plane8(SampleSrc* src, const QuantSet* q, VlcState* vs, uint8_t* rst, uint8_t* dst, int stride, int w, int h) {
    const int LW = w + 6;
    int16_t* lines = (int16_t*)calloc((size_t)3 * LW, sizeof(int16_t));
    if (!lines) return -23;
    int run_index = 0, err = 0;
    for (int y = 0; y < h && !err; ++y) {
        int16_t *cur = lines + (size_t)((y + 3) % 3) * LW + 3, *last = lines + (size_t)((y + 2) % 3) * LW + 3, *last2 = lines + (size_t)((y + 1) % 3) * LW + 3;
        if (y == 0) { memset(last - 3, 0, sizeof(int16_t) * LW); memset(last2 - 3, 0, sizeof(int16_t) * LW); }
        else if (y == 1) memset(last2 - 3, 0, sizeof(int16_t) * LW);
        cur[-1] = last[0]; last[w] = last[w - 1];
        cur[-2] = 0;
        int run_count = 0, run_mode = 0;
        for (int x = 0; x < w; ++x) {
            int context = get_context(q, cur + x, last + x, last2 + x), sign = 0, diff;
            if (context < 0) { context = -context; sign = 1; }
            if (context >= q->context_count) { err = 1; break; }
            if (src->ac) {
                diff = get_symbol(src->c, rst + (size_t)context * CONTEXT_SIZE, 1, &err);
                if (err) break;
            } else {
                if (context == 0 && run_mode == 0) run_mode = 1;
                if (run_mode) {
                    if (run_count == 0 && run_mode == 1) {
                        if (br_get1(src->br)) {
                            run_count = 1 << log2_run[run_index];
                            if (x + run_count <= w) run_index++;
                        } else {
                            run_count = log2_run[run_index] ? (int)br_get(src->br, log2_run[run_index]) : 0;
                            if (run_index) run_index--;
                            run_mode = 2;
                        }
                    }
                    run_count--;
                    if (run_count < 0) {
                        run_mode = 0; run_count = 0;
                        diff = get_vlc_symbol(src->br, &vs[context], 8);
                        if (diff >= 0) diff++;
                    } else {
                        diff = 0;
                    }
                } else {
                    diff = get_vlc_symbol(src->br, &vs[context], 8);
                }
            }
            if (sign) diff = -diff;
            cur[x] = (int16_t)((mid_pred(cur[x - 1], cur[x - 1] + last[x] - last[x - 1], last[x]) + diff) & 0xFF);
        }
        if (!err) for (int x = 0; x < w; ++x) dst[(size_t)y * stride + x] = (uint8_t)cur[x];
    }
    free(lines);
    return err ? -25 : 0;
}

/* adaptive context states of one slice, kept from frame to frame (a key frame resets them): one array per plane index (0 luma, 1 both chroma
 * planes, 2 alpha); Golomb-Rice streams keep VlcStates, range-coded streams CONTEXT_SIZE bytes per context */
typedef struct { VlcState* vlc[3]; uint8_t* rst[3]; int n[3]; int ac; int live; } SliceCtx;
#define MAX_SLICES 1024
typedef struct {
    Config cf;
    int have_cf;                       /* configuration known (version 3: from the record; versions 0 / 1: from the last key frame's header) */
    int legacy;                        /* versions 0 / 1 */
    int key_ok;                        /* a key frame has been decoded */
    SliceCtx sl[MAX_SLICES];
} Decoder;

static void slice_ctx_free(SliceCtx* s) {
    for (int p = 0; p < 3; ++p) { free(s->vlc[p]); free(s->rst[p]); s->vlc[p] = 0; s->rst[p] = 0; s->n[p] = 0; }
    s->live = 0;
}
/* key frame: (re)allocate and reset; non-key frame: the states of the previous frame must be there with the same shape */
static int slice_ctx_prepare(SliceCtx* s, const QuantSet* const* qs, int ac, int keyframe) {
    if (!keyframe) {
        if (!s->live || s->ac != ac) return -20;
        for (int p = 0; p < 3; ++p) if (s->n[p] != qs[p]->context_count) return -20;
        return 0;
    }
    for (int p = 0; p < 3; ++p) {
        const int n = qs[p]->context_count;
        if (n < 1 || n > MAX_CTX) return -21;
        if (s->n[p] != n || s->ac != ac || !s->live) {
            free(s->vlc[p]); free(s->rst[p]); s->vlc[p] = 0; s->rst[p] = 0; s->n[p] = 0;
            if (ac) s->rst[p] = (uint8_t*)malloc((size_t)n * CONTEXT_SIZE);
            else s->vlc[p] = (VlcState*)malloc(sizeof(VlcState) * (size_t)n);
            if (!s->rst[p] && !s->vlc[p]) { s->live = 0; return -23; }
            s->n[p] = n;
        }
        if (ac) memset(s->rst[p], 128, (size_t)n * CONTEXT_SIZE);
        else for (int i = 0; i < n; ++i) { s->vlc[p][i].drift = 0; s->vlc[p][i].error_sum = 4; s->vlc[p][i].bias = 0; s->vlc[p][i].count = 1; }
    }
    s->ac = ac; s->live = 1;
    return 0;
}

/* the sample data of one slice.  c: the slice's range coder, positioned behind everything that precedes the samples (key-frame bit, frame
 * header of versions 0 / 1, slice header of version 3); data / len: the slice's bytes; (x0, y0, w, h): its rectangle; qi: quantisation-table
 * set per plane index */
static int decode_slice_samples(const Config* cf, SliceCtx* sc, RangeCoder* pc, const uint8_t* data, int len, int x0, int y0, int w, int h, const int* qi,
                                int keyframe, int W, int H, uint8_t* rgb, uint8_t* const* yuv) {
    RangeCoder c = *pc;
    int err = 0;
    const int ac = cf->coder != 0;                     /* sample data range coded (coder_type 1 / 2) instead of Golomb-Rice */
    const int nplanes = 3 + (cf->alpha ? 1 : 0);       /* G, B-G, R-G (JPEG 2000 RCT) [, alpha: decoded, dropped] */
    BitR br = {data, 0, 0, 0};
    if (!ac) {
        if (cf->version == 3 && cf->micro > 1) { uint8_t st = 129; (void)get_rac(&c, &st); }
        const int ac_bytes = (int)(c.ptr - c.start) - 1;
        if (ac_bytes < 0 || ac_bytes > len) return -22;
        br.buf = data + ac_bytes; br.nbits = (int64_t)(len - ac_bytes) * 8;
    }
    const QuantSet* qs[3] = {&cf->sets[qi[0]], &cf->sets[qi[1]], &cf->sets[qi[2]]};
    int r = slice_ctx_prepare(sc, qs, ac, keyframe);
    if (r < 0) return r;
    VlcState** vlc = sc->vlc;
    uint8_t** rst = sc->rst;
    if (cf->colorspace == 0) {
        /* planar YCbCr: Y (index 0), then Cb and Cr (index 1, shared states) on the subsampled grid, then alpha (index 2; decoded, dropped) */
        SampleSrc src = {ac, &c, &br};
        const int cw = cf->chroma_planes ? (w + (1 << cf->hshift) - 1) >> cf->hshift : 0, chh = cf->chroma_planes ? (h + (1 << cf->vshift) - 1) >> cf->vshift : 0;
        const int cx = x0 >> cf->hshift, cy = y0 >> cf->vshift;
        const int CW = (W + (1 << cf->hshift) - 1) >> cf->hshift, CH = (H + (1 << cf->vshift) - 1) >> cf->vshift;
        r = decode_plane8(&src, qs[0], vlc[0], rst[0], yuv[0] + (size_t)y0 * W + x0, W, w, h);
        if (!r && cf->chroma_planes) {
            if (cx + cw > CW || cy + chh > CH) r = -21;
            if (!r) r = decode_plane8(&src, qs[1], vlc[1], rst[1], yuv[1] + (size_t)cy * CW + cx, CW, cw, chh);
            if (!r) r = decode_plane8(&src, qs[1], vlc[1], rst[1], yuv[2] + (size_t)cy * CW + cx, CW, cw, chh);
        }
        if (!r && cf->alpha) {
            uint8_t* scratch = (uint8_t*)malloc((size_t)w * h);
            if (!scratch) r = -23;
            else { r = decode_plane8(&src, qs[2], vlc[2], rst[2], scratch, w, w, h); free(scratch); }
        }
        if (r) return r;
        if (ac) return c.overflow > 2 ? -24 : 0;
        return br.overflow ? -24 : 0;
    }
    const int LW = w + 6;
    int16_t* lines = (int16_t*)calloc((size_t)4 * 3 * LW, sizeof(int16_t));
    if (!lines) return -23;
    int run_index = 0;
    for (int y = 0; y < h && !err; ++y) {
        int16_t* s[4][3];
        for (int p = 0; p < 4; ++p)
            for (int i = 0; i < 3; ++i) s[p][i] = lines + ((size_t)p * 3 + (size_t)((y + 3 - i) % 3)) * LW + 3;
        for (int p = 0; p < nplanes; ++p) {
            int16_t *cur = s[p][0], *last = s[p][1], *last2 = s[p][2];
            const int set = (p + 1) / 2;               /* plane -> plane index: 0, 1, 1, 2 */
            const QuantSet* q = qs[set];
            if (y == 0) { memset(last - 3, 0, sizeof(int16_t) * LW); memset(last2 - 3, 0, sizeof(int16_t) * LW); }
            else if (y == 1) memset(last2 - 3, 0, sizeof(int16_t) * LW);
            cur[-1] = last[0]; last[w] = last[w - 1];
            cur[-2] = 0;                               /* RFC 9043 3.2 (border): one left column = the row above shifted down, the ADDITIONAL left column is 0 */
            int run_count = 0, run_mode = 0;
            for (int x = 0; x < w; ++x) {
                int context = get_context(q, cur + x, last + x, last2 + x), sign = 0, diff;
                if (context < 0) { context = -context; sign = 1; }
                if (context >= q->context_count) { err = 1; break; }      /* inconsistent (crafted) quantisation tables */
                if (ac) {
                    diff = get_symbol(&c, rst[set] + (size_t)context * CONTEXT_SIZE, 1, &err);
                    if (err) break;
                } else {
                    VlcState* vs = vlc[set];
                    if (context == 0 && run_mode == 0) run_mode = 1;
                    if (run_mode) {
                        if (run_count == 0 && run_mode == 1) {
                            if (br_get1(&br)) {
                                run_count = 1 << log2_run[run_index];
                                if (x + run_count <= w) run_index++;
                            } else {
                                run_count = log2_run[run_index] ? (int)br_get(&br, log2_run[run_index]) : 0;
                                if (run_index) run_index--;
                                run_mode = 2;
                            }
                        }
                        run_count--;
                        if (run_count < 0) {
                            run_mode = 0; run_count = 0;
                            diff = get_vlc_symbol(&br, &vs[context], 9);
                            if (diff >= 0) diff++;
                        } else {
                            diff = 0;
                        }
                    } else {
                        diff = get_vlc_symbol(&br, &vs[context], 9);
                    }
                }
                if (sign) diff = -diff;
                cur[x] = (int16_t)((mid_pred(cur[x - 1], cur[x - 1] + last[x] - last[x - 1], last[x]) + diff) & 0x1FF);
            }
            if (err) break;
        }
        if (err) break;
        uint8_t* row = rgb + ((size_t)(y0 + y) * W + x0) * 3;
        for (int x = 0; x < w; ++x) {
            int g = s[0][0][x], b = s[1][0][x], r2 = s[2][0][x];
            b -= 256; r2 -= 256; g -= (b + r2) >> 2; b += g; r2 += g;
            row[3 * x] = (uint8_t)r2; row[3 * x + 1] = (uint8_t)g; row[3 * x + 2] = (uint8_t)b;
        }
    }
    free(lines);
    if (err) return -25;
    if (ac) return c.overflow > 2 ? -24 : 0;           /* the range decoder legitimately reads up to two bytes past the coded data */
    return br.overflow ? -24 : 0;
}

static void apply_custom_table(const Config* cf, RangeCoder* c) {     /* coder_type 2 (libavcodec ff_ffv1_init_slice_state) */
    if (cf->coder == 2)
        for (int i = 1; i < 256; ++i) { c->one_state[i] = cf->one_state[i]; c->zero_state[256 - i] = (uint8_t)(256 - cf->one_state[i]); }
}

/* one slice of a version-3 packet: [key-frame bit in the first slice] slice header, samples */
static int decode_slice_v3(const Config* cf, SliceCtx* sc, const uint8_t* data, int len, int first, int* keyframe, int key_ok, int W, int H,
                           uint8_t* rgb, uint8_t* const* yuv) {
    RangeCoder c;
    uint8_t state[CONTEXT_SIZE];
    int err = 0;
    const int nqi = 1 + (cf->chroma_planes ? 1 : 0) + (cf->alpha ? 1 : 0);
    rc_init_dec(&c, data, len);
    if (first) {
        uint8_t keystate = 128;
        *keyframe = get_rac(&c, &keystate);
        if (!*keyframe && !key_ok) return -20;         /* a stream cannot start with a non-key frame */
    }
    apply_custom_table(cf, &c);
    memset(state, 128, sizeof(state));
    const int sx = get_symbol(&c, state, 0, &err), sy = get_symbol(&c, state, 0, &err);
    const int sw = get_symbol(&c, state, 0, &err) + 1, sh = get_symbol(&c, state, 0, &err) + 1;
    int qi[3] = {0, 0, 0};
    for (int i = 0; i < nqi; ++i) qi[i] = get_symbol(&c, state, 0, &err);
    if (!cf->chroma_planes) { qi[2] = qi[1]; qi[1] = qi[0]; }
    (void)get_symbol(&c, state, 0, &err); (void)get_symbol(&c, state, 0, &err); (void)get_symbol(&c, state, 0, &err);
    /* every field came from the file: refuse anything outside the slice grid / the table sets before it indexes memory */
    if (err || sx < 0 || sy < 0 || sw < 1 || sh < 1 || sx > cf->nh - sw || sy > cf->nv - sh) return -21;
    for (int i = 0; i < 3; ++i) if (qi[i] < 0 || qi[i] >= cf->nsets) return -21;
    const int y0 = (int)((int64_t)sy * H / cf->nv), h = (int)((int64_t)(sy + sh) * H / cf->nv) - y0;
    const int x0 = (int)((int64_t)sx * W / cf->nh), w = (int)((int64_t)(sx + sw) * W / cf->nh) - x0;
    if (w < 1 || h < 1 || x0 + w > W || y0 + h > H) return -21;
    return decode_slice_samples(cf, sc, &c, data, len, x0, y0, w, h, qi, *keyframe, W, H, rgb, yuv);
}

/* versions 0 / 1: the frame header (libavcodec ffv1dec.c read_header, version < 2) behind the key-frame bit of a key frame */
static int parse_legacy_header(RangeCoder* c, Config* cf) {
    uint8_t state[CONTEXT_SIZE];
    int err = 0;
    memset(cf, 0, sizeof(*cf));
    memset(state, 128, sizeof(state));
    cf->version = get_symbol(c, state, 0, &err);
    if (err || cf->version < 0 || cf->version > 1) return -2;
    cf->coder = get_symbol(c, state, 0, &err);
    if (err || cf->coder < 0 || cf->coder > 2) return -3;
    if (cf->coder == 2) {
        for (int i = 1; i < 256; ++i) {
            const int v = get_symbol(c, state, 1, &err) + c->one_state[i];
            if (err || v < 1 || v > 255) return -3;
            cf->one_state[i] = (uint8_t)v;
        }
    }
    cf->colorspace = get_symbol(c, state, 0, &err);
    cf->bits = cf->version > 0 ? get_symbol(c, state, 0, &err) : 8;
    cf->chroma_planes = get_rac(c, state);
    cf->hshift = get_symbol(c, state, 0, &err);
    cf->vshift = get_symbol(c, state, 0, &err);
    cf->alpha = get_rac(c, state);
    cf->nh = cf->nv = 1; cf->nsets = 1; cf->ec = 0; cf->intra = 0;
    int count = 1;
    for (int t = 0; t < 5; ++t) {
        const int r = read_quant_table(c, cf->sets[0].quant[t], count);
        if (r < 0) return -5;
        count *= r;
        if (count > MAX_CTX) return -5;
    }
    cf->sets[0].context_count = (count + 1) / 2;
    cf->sets[0].five = cf->sets[0].quant[3][127] || cf->sets[0].quant[4][127];
    if (err) return -7;
    if ((cf->colorspace != 0 && cf->colorspace != 1) || (cf->bits != 0 && cf->bits != 8)) return -8;
    if (cf->colorspace == 0 && (cf->hshift < 0 || cf->hshift > 2 || cf->vshift < 0 || cf->vshift > 2 || (cf->alpha && !cf->chroma_planes))) return -8;
    return 0;
}

static int decoder_decode(Decoder* d, const uint8_t* data, int len, int W, int H, uint8_t* rgb, uint8_t* const* yuv) {
    int r;
    if (len < 2 || W < 1 || H < 1) return -11;
    if (d->legacy) {
        /* versions 0 / 1: ONE range coder for the key-frame bit, the header (key frames) and -- range-coded streams -- the samples */
        RangeCoder c;
        uint8_t keystate = 128;
        rc_init_dec(&c, data, len);
        const int keyframe = get_rac(&c, &keystate);
        if (keyframe) {
            r = parse_legacy_header(&c, &d->cf);
            if (r < 0) { d->have_cf = 0; d->key_ok = 0; return r; }
            d->have_cf = 1;
        } else if (!d->key_ok || !d->have_cf) {
            return -20;
        }
        const Config* cf = &d->cf;
        if (cf->colorspace == 0 ? !yuv : !rgb) return -9;
        if (cf->colorspace == 0 && !cf->chroma_planes) {
            const size_t n = (size_t)((W + (1 << cf->hshift) - 1) >> cf->hshift) * (size_t)((H + (1 << cf->vshift) - 1) >> cf->vshift);
            memset(yuv[1], 128, n); memset(yuv[2], 128, n);
        }
        apply_custom_table(cf, &c);
        const int qi[3] = {0, 0, 0};
        r = decode_slice_samples(cf, &d->sl[0], &c, data, len, 0, 0, W, H, qi, keyframe, W, H, rgb, yuv);
        d->key_ok = r == 0 ? 1 : 0;
        return r;
    }
    const Config* cf = &d->cf;
    if (cf->colorspace == 0 ? !yuv : !rgb) return -9;
    if (cf->colorspace == 0 && !cf->chroma_planes) {
        const size_t n = (size_t)((W + (1 << cf->hshift) - 1) >> cf->hshift) * (size_t)((H + (1 << cf->vshift) - 1) >> cf->vshift);
        memset(yuv[1], 128, n); memset(yuv[2], 128, n);
    }
    /* walk the slices from the END of the packet: [... slice][size:3][status:1 crc:4 when ec] */
    const int trailer = 3 + (cf->ec ? 5 : 0);
    int end = len, nslices = 0;
    static const int cap = MAX_SLICES;
    int* starts = (int*)malloc(sizeof(int) * 2 * (size_t)cap);
    if (!starts) return -23;
    int* lens = starts + cap;
    r = 0;
    while (end > 0) {
        if (end < trailer) { r = -11; break; }
        const uint8_t* t = data + end - trailer;
        const int size = (t[0] << 16) | (t[1] << 8) | t[2];
        const int total = size + trailer;
        if (total > end || nslices >= cap) { r = -12; break; }
        if (cf->ec && crc32_mpeg(data + end - total, total) != 0) { r = -13; break; }
        starts[nslices] = end - total; lens[nslices] = size; nslices++;
        end -= total;
    }
    int keyframe = 0;
    for (int i = nslices - 1, k = 0; !r && i >= 0; --i, ++k)
        r = decode_slice_v3(cf, &d->sl[k], data + starts[i], lens[i], k == 0, &keyframe, d->key_ok, W, H, rgb, yuv);
    free(starts);
    d->key_ok = r == 0 ? 1 : 0;
    return r;
}

/* ---- the stateful decoder object (non-key frames continue from the previous frame's context states) ------------------------------------- */
/* cfg / cfglen: the stream's configuration record (Matroska CodecPrivate) -- version 3; cfglen == 0: a version 0 / 1 stream (parameters in the
 * key frames).  *status = 0 or the error code.  Returns an opaque handle or NULL */
void* vvio_ffv1_decoder_open(const uint8_t* cfg, int cfglen, int* status) {
    Decoder* d = (Decoder*)calloc(1, sizeof(Decoder));
    int r = 0;
    if (!d) r = -1;
    else if (cfglen <= 0) d->legacy = 1;
    else if (cfglen < 5 || crc32_mpeg(cfg, cfglen) != 0) r = -10;
    else { r = parse_config(cfg, cfglen - 4, &d->cf); d->have_cf = r == 0; }
    if (r < 0) { free(d); d = 0; }
    if (status) *status = r;
    return d;
}
void vvio_ffv1_decoder_close(void* h) {
    Decoder* d = (Decoder*)h;
    if (!d) return;
    for (int i = 0; i < MAX_SLICES; ++i) slice_ctx_free(&d->sl[i]);
    free(d);
}
/* info[7] = colorspace_type (0 YCbCr, 1 RGB), chroma_planes, log2_h_chroma_subsample, log2_v_chroma_subsample, extra_plane, bits_per_raw_sample,
 * version; -30 while the parameters are not known yet (version 0 / 1 before the first key frame) */
int vvio_ffv1_decoder_info(void* h, int* info) {
    Decoder* d = (Decoder*)h;
    if (!d || !info) return -1;
    if (!d->have_cf) return -30;
    const Config* cf = &d->cf;
    info[0] = cf->colorspace; info[1] = cf->chroma_planes; info[2] = cf->hshift; info[3] = cf->vshift; info[4] = cf->alpha; info[5] = cf->bits ? cf->bits : 8;
    info[6] = cf->version;
    return 0;
}
/* the next packet of the stream, in stream order.  RGB streams fill rgb (W*H*3) and return 0; planar YCbCr streams fill y / cb / cr and
 * return 1 (gray streams: cb = cr = 128); negative = error code (both buffer kinds may be passed: the stream decides) */
int vvio_ffv1_decoder_decode(void* h, const uint8_t* data, int len, int W, int H, uint8_t* rgb, uint8_t* y, uint8_t* cb, uint8_t* cr) {
    Decoder* d = (Decoder*)h;
    if (!d || !data) return -1;
    uint8_t* planes[3] = {y, cb, cr};
    const int r = decoder_decode(d, data, len, W, H, rgb, (y && cb && cr) ? planes : 0);
    if (r < 0) return r;
    return d->cf.colorspace == 0 ? 1 : 0;
}

/* ---- stateless entry points: one KEY frame of a version-3 stream ---------------------------------------------------------------------------- */
static int decode_frame_any(const uint8_t* cfg, int cfglen, const uint8_t* data, int len, int W, int H, uint8_t* rgb, uint8_t* const* yuv) {
    int r = 0;
    if (cfglen <= 0) return -10;
    Decoder* d = (Decoder*)vvio_ffv1_decoder_open(cfg, cfglen, &r);
    if (!d) return r;
    if ((d->cf.colorspace == 0) != (yuv != 0)) { vvio_ffv1_decoder_close(d); return -9; }      /* the caller asked for the other colour model (vvio_ffv1_stream_info) */
    r = decoder_decode(d, data, len, W, H, rgb, yuv);
    vvio_ffv1_decoder_close(d);
    return r;
}

/* FFV1 packet + configuration record -> RGB24 (W*H*3 bytes).  0 = ok, negative = error code */
int vvio_ffv1_decode_frame(const uint8_t* cfg, int cfglen, const uint8_t* data, int len, int W, int H, uint8_t* rgb) {
    return decode_frame_any(cfg, cfglen, data, len, W, H, rgb, 0);
}

/* info[6] = colorspace_type (0 YCbCr, 1 RGB), chroma_planes, log2_h_chroma_subsample, log2_v_chroma_subsample, extra_plane, bits_per_raw_sample */
int vvio_ffv1_stream_info(const uint8_t* cfg, int cfglen, int* info) {
    Config* cf = (Config*)malloc(sizeof(Config));
    if (!cf || !info) { free(cf); return -1; }
    if (cfglen < 5 || crc32_mpeg(cfg, cfglen) != 0) { free(cf); return -10; }
    const int r = parse_config(cfg, cfglen - 4, cf);
    if (r == 0) { info[0] = cf->colorspace; info[1] = cf->chroma_planes; info[2] = cf->hshift; info[3] = cf->vshift; info[4] = cf->alpha; info[5] = cf->bits ? cf->bits : 8; }
    free(cf);
    return r;
}

/* planar 8-bit YCbCr stream (colorspace_type 0) -> Y [H][W], Cb / Cr [ceil(H >> vshift)][ceil(W >> hshift)] (gray streams: Cb = Cr = 128) */
int vvio_ffv1_decode_frame_yuv(const uint8_t* cfg, int cfglen, const uint8_t* data, int len, int W, int H, uint8_t* y, uint8_t* cb, uint8_t* cr) {
    if (!y || !cb || !cr) return -1;
    uint8_t* planes[3] = {y, cb, cr};
    return decode_frame_any(cfg, cfglen, data, len, W, H, 0, planes);
}

/* ---------------------------------------------------------------------------------------------------------- colour conversion */
/* YCbCr (ITU-R BT.601 matrix; limited range 16..235 / 16..240, or full range) -> RGB24 in integer arithmetic -- the SAME arithmetic as the HIP
 * kernel vv_ycbcr_to_rgb (videovanish_amd/csrc/vv_image.hip), bit for bit.  Chroma siting of MPEG-2 / H.264 4:2:0: co-sited with the even luma
 * columns, midway between two luma rows; bilinear chroma interpolation in 1/16 units (weights 4 | 2+2 horizontally, 1+3 | 3+1 vertically);
 * subsampling factors above 2 replicate.  What cv2.VideoCapture hands the reference (tools.py:17-21) comes out of libswscale (bicubic chroma,
 * its own fixed point): PARITY UNPINNED against it. */
static inline int clamp_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
static inline int chroma16(const uint8_t* c, int CW, int CH, int X, int Y, int hs, int vs) {
    int x0 = X >> hs, x1 = x0, wx0 = 4, wx1 = 0, y0 = Y >> vs, y1 = y0, wy0 = 4, wy1 = 0;
    if (hs == 1 && (X & 1)) { x1 = x0 + 1 < CW ? x0 + 1 : x0; wx0 = 2; wx1 = 2; }
    if (vs == 1) {
        if (Y & 1) { y1 = y0 + 1 < CH ? y0 + 1 : y0; wy0 = 3; wy1 = 1; }
        else { y1 = y0 > 0 ? y0 - 1 : y0; wy0 = 3; wy1 = 1; }
    }
    return wy0 * (wx0 * c[(size_t)y0 * CW + x0] + wx1 * c[(size_t)y0 * CW + x1]) + wy1 * (wx0 * c[(size_t)y1 * CW + x0] + wx1 * c[(size_t)y1 * CW + x1]);
}
int vvio_ycbcr_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int W, int H, int hshift, int vshift, int full_range, uint8_t* rgb) {
    if (!y || !cb || !cr || !rgb || W < 1 || H < 1 || hshift < 0 || hshift > 2 || vshift < 0 || vshift > 2) return -1;
    const int CW = (W + (1 << hshift) - 1) >> hshift, CH = (H + (1 << vshift) - 1) >> vshift;
    const int ky = full_range ? 65536 : 76309, yoff = full_range ? 0 : 16;
    const int krv = full_range ? 91881 : 104597, kgu = full_range ? 22554 : 25675, kgv = full_range ? 46802 : 53279, kbu = full_range ? 116130 : 132201;
    for (int Y = 0; Y < H; ++Y)
        for (int X = 0; X < W; ++X) {
            const int yy = 16 * ky * ((int)y[(size_t)Y * W + X] - yoff);
            const int u = chroma16(cb, CW, CH, X, Y, hshift, vshift) - 2048, v = chroma16(cr, CW, CH, X, Y, hshift, vshift) - 2048;
            uint8_t* o = rgb + ((size_t)Y * W + X) * 3;
            o[0] = (uint8_t)clamp_u8((yy + krv * v + (1 << 19)) >> 20);
            o[1] = (uint8_t)clamp_u8((yy - kgu * u - kgv * v + (1 << 19)) >> 20);
            o[2] = (uint8_t)clamp_u8((yy + kbu * u + (1 << 19)) >> 20);
        }
    return 0;
}

int vvio_abi_version(void) { return 3; }
