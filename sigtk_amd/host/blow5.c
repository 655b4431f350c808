/* blow5.c -- see blow5.h */
#include "blow5.h"

#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <zlib.h>

/* an inflated record larger than this is refused (a 2^31-sample record is 4 GiB of signal) */
#define B5_MAX_INFLATED (1ull << 33)

static const uint8_t B5_MAGIC[6] = {'B', 'L', 'O', 'W', '5', 1};
static const char B5_EOF_MARK[5] = {'5', 'W', 'O', 'L', 'B'};

static int grow(uint8_t **p, uint64_t *cap, uint64_t need) {
    if (*cap >= need) return 0;
    uint64_t c = *cap ? *cap : 512;
    while (c < need) c *= 2;
    uint8_t *q = (uint8_t *)realloc(*p, c);
    if (!q) return B5_ERR_MEM;
    *p = q;
    *cap = c;
    return 0;
}

b5_file_t *b5_open(const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    uint8_t head[68];
    if (fread(head, 1, 68, fp) != 68 || memcmp(head, B5_MAGIC, 6) != 0) {
        fclose(fp);
        return NULL;
    }
    b5_file_t *f = (b5_file_t *)calloc(1, sizeof *f);
    if (!f) { fclose(fp); return NULL; }
    f->fp = fp;
    f->path = strdup(path);
    memcpy(f->version, head + 6, 3);
    f->record_press = head[9];
    memcpy(&f->num_read_groups, head + 10, 4);
    /* the signal compression byte exists from file version 0.2.0 on */
    const int has_sig = f->version[0] > 0 || f->version[1] >= 2;
    f->signal_press = has_sig ? head[14] : 0;
    uint32_t hsize;
    memcpy(&hsize, head + 64, 4);
    f->hdr_text = (char *)malloc((size_t)hsize + 1);
    if (!f->hdr_text || fread(f->hdr_text, 1, hsize, fp) != hsize) { b5_close(f); return NULL; }
    f->hdr_text[hsize] = '\0';
    f->hdr_size = hsize;
    f->first_rec = 68 + (uint64_t)hsize;
    if (f->record_press > 1 || f->signal_press > 1) { b5_close(f); return NULL; }
    return f;
}

void b5_close(b5_file_t *f) {
    if (!f) return;
    if (f->map) munmap((void *)f->map, (size_t)f->map_len);
    if (f->fp) fclose(f->fp);
    for (uint64_t i = 0; i < f->n_idx; i++) free(f->idx[i].id);
    free(f->idx);
    free(f->hdr_text);
    free(f->path);
    free(f);
}

char *b5_hdr_get(const b5_file_t *f, const char *name, uint32_t rg) {
    const size_t nl = strlen(name);
    const char *p = f->hdr_text;
    while (p && *p) {
        const char *eol = strchr(p, '\n');
        const size_t len = eol ? (size_t)(eol - p) : strlen(p);
        if (len > nl + 1 && p[0] == '@' && strncmp(p + 1, name, nl) == 0 && p[1 + nl] == '\t') {
            const char *v = p + 2 + nl, *end = p + len;
            for (uint32_t g = 0; g < rg; g++) {
                const char *t = memchr(v, '\t', (size_t)(end - v));
                if (!t) return NULL;
                v = t + 1;
            }
            const char *t = memchr(v, '\t', (size_t)(end - v));
            const size_t vl = t ? (size_t)(t - v) : (size_t)(end - v);
            char *out = (char *)malloc(vl + 1);
            if (!out) return NULL;
            memcpy(out, v, vl);
            out[vl] = '\0';
            return out;
        }
        p = eol ? eol + 1 : NULL;
    }
    return NULL;
}

/* streamvbyte (scalar) + zigzag-delta, slow5lib/src/slow5_press.c:1116-1146: u32 count, 2-bit
 * length codes (four per key byte, LSB first), then the little-endian data bytes */
static int svb_zd_decode(const uint8_t *blob, uint64_t nbytes, b5_rec_t *rec) {
    if (nbytes < 4) return B5_ERR_PRESS;
    uint32_t count;
    memcpy(&count, blob, 4);
    const uint64_t nkeys = ((uint64_t)count + 3) / 4;
    /* every value takes a key and at least one data byte: a short blob cannot claim a huge count */
    if (count > 0x7fffffffu || 4 + nkeys + (uint64_t)count > nbytes) return B5_ERR_PRESS;
    if (rec->cap_signal < count) {
        int16_t *q = (int16_t *)realloc(rec->raw_signal, sizeof(int16_t) * ((uint64_t)count + 1));
        if (!q) return B5_ERR_MEM;
        rec->raw_signal = q;
        rec->cap_signal = count;
    }
    const uint8_t *keys = blob + 4, *data = keys + nkeys, *end = blob + nbytes;
    int32_t prev = 0;
    for (uint32_t i = 0; i < count; i++) {
        const unsigned code = (keys[i >> 2] >> ((i & 3) * 2)) & 3u;
        if (data + code + 1 > end) return B5_ERR_PRESS;
        uint32_t v = 0;
        memcpy(&v, data, code + 1);
        data += code + 1;
        const int32_t delta = (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
        prev += delta;
        rec->raw_signal[i] = (int16_t)prev;
    }
    if (data != end) return B5_ERR_PRESS;
    rec->len_raw_signal = count;
    return 0;
}

static int parse_record(const b5_file_t *f, const uint8_t *p, uint64_t n, b5_rec_t *rec, int id_only) {
    if (n < 2) return B5_ERR_FORMAT;
    uint16_t idl;
    memcpy(&idl, p, 2);
    if (2 + (uint64_t)idl + 44 > n) return B5_ERR_FORMAT;
    char *id = (char *)realloc(rec->read_id, (size_t)idl + 1);
    if (!id) return B5_ERR_MEM;
    memcpy(id, p + 2, idl);
    id[idl] = '\0';
    rec->read_id = id;
    if (id_only) return 0;
    const uint8_t *q = p + 2 + idl;
    memcpy(&rec->read_group, q, 4);
    memcpy(&rec->digitisation, q + 4, 8);
    memcpy(&rec->offset, q + 12, 8);
    memcpy(&rec->range, q + 20, 8);
    memcpy(&rec->sampling_rate, q + 28, 8);
    uint64_t ln;
    memcpy(&ln, q + 36, 8);
    q += 44;
    const uint64_t left = n - (uint64_t)(q - p);
    if (f->signal_press == 1) {
        if (ln > left) return B5_ERR_FORMAT;
        return svb_zd_decode(q, ln, rec);
    }
    if (ln > 0x7fffffffull || ln * 2 > left) return B5_ERR_FORMAT; /* ln is untrusted: no wrap in ln * 2 */
    if (rec->cap_signal < ln) {
        int16_t *s = (int16_t *)realloc(rec->raw_signal, sizeof(int16_t) * (ln + 1));
        if (!s) return B5_ERR_MEM;
        rec->raw_signal = s;
        rec->cap_signal = ln;
    }
    memcpy(rec->raw_signal, q, ln * 2);
    rec->len_raw_signal = ln;
    return 0;
}

/* reads the record at the current file position; id_only skips signal decoding when possible */
static int read_record(b5_file_t *f, b5_rec_t *rec, int id_only) {
    uint8_t szb[8];
    const size_t got = fread(szb, 1, 8, f->fp);
    if (got >= 5 && memcmp(szb, B5_EOF_MARK, 5) == 0) {
        /* proper end only if nothing follows the marker */
        if (got == 5) return B5_EOF;
        return B5_ERR_FORMAT;
    }
    if (got != 8) return B5_ERR_IO;
    uint64_t size;
    memcpy(&size, szb, 8);
    if (size == 0 || size > (1ull << 36)) return B5_ERR_FORMAT;
    int rc = grow(&rec->buf, &rec->cap_buf, size);
    if (rc) return rc;
    if (fread(rec->buf, 1, size, f->fp) != size) return B5_ERR_IO;
    const uint8_t *p = rec->buf;
    uint64_t n = size;
    if (f->record_press == 1) {
        uLongf cap = rec->cap_zbuf ? rec->cap_zbuf : (size * 4 + 4096);
        for (;;) {
            rc = grow(&rec->zbuf, &rec->cap_zbuf, cap);
            if (rc) return rc;
            uLongf out = rec->cap_zbuf;
            const int z = uncompress(rec->zbuf, &out, rec->buf, size);
            if (z == Z_OK) { n = out; break; }
            if (z != Z_BUF_ERROR) return B5_ERR_PRESS;
            if (rec->cap_zbuf >= B5_MAX_INFLATED) return B5_ERR_PRESS; /* zip bomb / truncated stream */
            cap = rec->cap_zbuf * 2;
        }
        p = rec->zbuf;
    }
    return parse_record(f, p, n, rec, id_only);
}

int b5_next(b5_file_t *f, b5_rec_t *rec) { return read_record(f, rec, 0); }

static int idx_cmp(const void *a, const void *b) {
    return strcmp(((const b5_idx_entry_t *)a)->id, ((const b5_idx_entry_t *)b)->id);
}

/* ---- the on-disk index "<file>.idx" the reference keeps beside a BLOW5 (slow5lib/src/slow5_idx.c:360-500;
 * src/cmain.c:127-131 loads or creates it before reading by id): magic "SLOW5IDX\1", the file's version (3 x u8),
 * zeros up to offset 64, then (u16 id length, id, u64 offset of the record's size field, u64 size = 8 + record
 * bytes) per record, then "XDI5WOLS".  Reused when present and consistent, written (best effort) after a scan. */
static const char B5_IDX_MAGIC[9] = {'S', 'L', 'O', 'W', '5', 'I', 'D', 'X', '\1'};
static const char B5_IDX_EOF[8] = {'X', 'D', 'I', '5', 'W', 'O', 'L', 'S'};

static char *idx_path(const b5_file_t *f) {
    const size_t n = strlen(f->path);
    char *p = (char *)malloc(n + 5);
    if (!p) return NULL;
    memcpy(p, f->path, n);
    memcpy(p + n, ".idx", 5);
    return p;
}

/* 0: loaded into f->idx; non-zero: absent / stale / malformed (the caller scans the file instead) */
static int idx_load(b5_file_t *f) {
    char *ip = idx_path(f);
    if (!ip) return B5_ERR_MEM;
    /* an index older than its BLOW5 may describe another file of that name.  Like slow5lib (slow5_idx.c:43) this warns
     * and goes on -- after cp / rsync without -t, tar extraction, or in a read-only directory a rescan of a large file on
     * every invocation costs minutes -- and relies on the check of every fetched record (b5_get / b5_get_raw: id and
     * size; a mismatch drops the index and scans the file) */
    struct stat st_idx, st_dat;
    if (stat(ip, &st_idx) == 0 && stat(f->path, &st_dat) == 0 &&
        (st_idx.st_mtim.tv_sec < st_dat.st_mtim.tv_sec ||
         (st_idx.st_mtim.tv_sec == st_dat.st_mtim.tv_sec && st_idx.st_mtim.tv_nsec < st_dat.st_mtim.tv_nsec)))
        fprintf(stderr, "[b5_idx::WARNING] Index file '%s' is older than its BLOW5: using it, every fetched record is "
                        "checked against it\n", ip);
    FILE *fp = fopen(ip, "rb");
    free(ip);
    if (!fp) return B5_ERR_NOTFOUND;
    uint8_t head[64];
    int rc = B5_ERR_FORMAT;
    b5_idx_entry_t *idx = NULL;
    uint64_t n = 0, cap = 0;
    if (fread(head, 1, 64, fp) != 64 || memcmp(head, B5_IDX_MAGIC, 9) != 0 || memcmp(head + 9, f->version, 3) != 0)
        goto done;
    if (fseek(f->fp, 0, SEEK_END) != 0 || fseek(fp, 0, SEEK_END) != 0) { rc = B5_ERR_IO; goto done; }
    const uint64_t fsize = (uint64_t)ftell(f->fp);
    const uint64_t isize = (uint64_t)ftell(fp);
    if (isize < 72 || fseek(fp, 64, SEEK_SET) != 0) goto done;
    uint64_t pos = 64;
    while (pos < isize - 8) {
        uint16_t idl;
        if (fread(&idl, 2, 1, fp) != 1 || pos + 2 + (uint64_t)idl + 16 > isize - 8) goto done;
        char *id = (char *)malloc((size_t)idl + 1);
        if (!id) { rc = B5_ERR_MEM; goto done; }
        uint64_t off_size[2];
        /* (offset and size are untrusted 64-bit values: compared without forming their sum) */
        if (fread(id, 1, idl, fp) != idl || fread(off_size, 8, 2, fp) != 2 || off_size[0] < f->first_rec ||
            off_size[1] < 8 || off_size[0] > fsize || off_size[1] > fsize - off_size[0]) {
            free(id);
            goto done;
        }
        id[idl] = '\0';
        if (n == cap) {
            cap = cap ? cap * 2 : 1024;
            b5_idx_entry_t *q = (b5_idx_entry_t *)realloc(idx, cap * sizeof *idx);
            if (!q) { free(id); rc = B5_ERR_MEM; goto done; }
            idx = q;
        }
        idx[n].id = id;
        idx[n].offset = off_size[0];
        idx[n].size = off_size[1];
        n++;
        pos += 2 + (uint64_t)idl + 16;
    }
    {
        uint8_t tail[8];
        if (pos == isize - 8 && fread(tail, 1, 8, fp) == 8 && memcmp(tail, B5_IDX_EOF, 8) == 0) rc = 0;
    }
done:
    fclose(fp);
    if (rc) {
        for (uint64_t i = 0; i < n; i++) free(idx[i].id);
        free(idx);
        return rc;
    }
    qsort(idx, n, sizeof *idx, idx_cmp);
    f->idx = idx;
    f->n_idx = n;
    return 0;
}

/* best effort: a read-only directory just means the next run scans again */
static void idx_write(const b5_file_t *f, const b5_idx_entry_t *in_file_order, uint64_t n) {
    char *ip = idx_path(f);
    if (!ip) return;
    FILE *fp = fopen(ip, "wb");
    if (!fp) fprintf(stderr, "[b5_idx::WARNING] cannot write index file '%s': the next run scans the file again\n", ip);
    if (fp) {
        uint8_t head[64];
        memset(head, 0, sizeof head);
        memcpy(head, B5_IDX_MAGIC, 9);
        memcpy(head + 9, f->version, 3);
        int ok = fwrite(head, 1, 64, fp) == 64;
        for (uint64_t i = 0; ok && i < n; i++) {
            const size_t l = strlen(in_file_order[i].id);
            const uint16_t idl = (uint16_t)l;
            ok = l <= 0xffff && fwrite(&idl, 2, 1, fp) == 1 && fwrite(in_file_order[i].id, 1, l, fp) == l &&
                 fwrite(&in_file_order[i].offset, 8, 1, fp) == 1 && fwrite(&in_file_order[i].size, 8, 1, fp) == 1;
        }
        ok = ok && fwrite(B5_IDX_EOF, 1, 8, fp) == 8;
        if (fclose(fp) != 0) ok = 0;
        if (!ok) remove(ip);
    }
    free(ip);
}

static void idx_drop(b5_file_t *f) {
    for (uint64_t i = 0; i < f->n_idx; i++) free(f->idx[i].id);
    free(f->idx);
    f->idx = NULL;
    f->n_idx = 0;
}

static int index_build(b5_file_t *f, int trust_disk) {
    if (f->idx) return 0;
    const long keep = ftell(f->fp);
    if (trust_disk && idx_load(f) == 0) {
        f->idx_from_disk = 1;
        fseek(f->fp, keep, SEEK_SET);
        return 0;
    }
    f->idx_from_disk = 0;
    if (fseek(f->fp, (long)f->first_rec, SEEK_SET) != 0) return B5_ERR_IO;
    b5_rec_t tmp;
    memset(&tmp, 0, sizeof tmp);
    uint64_t cap = 1024, n = 0;
    b5_idx_entry_t *idx = (b5_idx_entry_t *)malloc(cap * sizeof *idx);
    if (!idx) return B5_ERR_MEM;
    int rc;
    for (;;) {
        const uint64_t pos = (uint64_t)ftell(f->fp);
        rc = read_record(f, &tmp, 1);
        if (rc == B5_EOF) { rc = 0; break; }
        if (rc) break;
        if (n == cap) {
            cap *= 2;
            b5_idx_entry_t *q = (b5_idx_entry_t *)realloc(idx, cap * sizeof *idx);
            if (!q) { rc = B5_ERR_MEM; break; }
            idx = q;
        }
        idx[n].id = strdup(tmp.read_id);
        idx[n].offset = pos;
        idx[n].size = (uint64_t)ftell(f->fp) - pos;
        n++;
    }
    b5_rec_free(&tmp);
    if (rc) {
        for (uint64_t i = 0; i < n; i++) free(idx[i].id);
        free(idx);
        return rc;
    }
    idx_write(f, idx, n);  /* file order, as the reference writes it */
    qsort(idx, n, sizeof *idx, idx_cmp);
    f->idx = idx;
    f->n_idx = n;
    fseek(f->fp, keep, SEEK_SET);
    return 0;
}

int b5_index(b5_file_t *f) { return index_build(f, 1); }

/* An index loaded from disk is only as good as the file it was written for: what it points at must be the record it
 * names, of the size it says.  On a mismatch the index is dropped, the file scanned, the lookup repeated once. */
int b5_get(b5_file_t *f, const char *read_id, b5_rec_t *rec) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!f->idx) {
            const int rc = index_build(f, attempt == 0);
            if (rc) return rc;
        }
        b5_idx_entry_t key;
        key.id = (char *)read_id;
        key.offset = 0;
        const b5_idx_entry_t *e = (const b5_idx_entry_t *)bsearch(&key, f->idx, f->n_idx, sizeof key, idx_cmp);
        if (!e) {
            if (f->idx_from_disk && attempt == 0) { idx_drop(f); continue; }
            return B5_ERR_NOTFOUND;
        }
        const uint64_t esize = e->size;
        if (fseek(f->fp, (long)e->offset, SEEK_SET) != 0) return B5_ERR_IO;
        const int rc = read_record(f, rec, 0);
        const int fits = rc == 0 && rec->read_id && strcmp(rec->read_id, read_id) == 0 &&
                         (uint64_t)ftell(f->fp) - e->offset == esize;
        if (fits || !f->idx_from_disk || attempt == 1) return rc == 0 && !fits ? B5_ERR_FORMAT : rc;
        idx_drop(f);
    }
    return B5_ERR_FORMAT;
}

void b5_rec_free(b5_rec_t *rec) {
    free(rec->read_id);
    free(rec->raw_signal);
    free(rec->buf);
    free(rec->zbuf);
    memset(rec, 0, sizeof *rec);
}

/* ------------------------------------------------------------------ split API (pipelined reader) */

static int read_raw_here(b5_file_t *f, uint8_t **buf, uint64_t *len, uint64_t *cap, uint64_t *size) {
    uint8_t szb[8];
    const size_t got = fread(szb, 1, 8, f->fp);
    if (got >= 5 && memcmp(szb, B5_EOF_MARK, 5) == 0) return got == 5 ? B5_EOF : B5_ERR_FORMAT;
    if (got != 8) return B5_ERR_IO;
    uint64_t sz;
    memcpy(&sz, szb, 8);
    if (sz == 0 || sz > (1ull << 36)) return B5_ERR_FORMAT;
    const int rc = grow(buf, cap, *len + sz);
    if (rc) return rc;
    if (fread(*buf + *len, 1, sz, f->fp) != sz) return B5_ERR_IO;
    *len += sz;
    *size = sz;
    return 0;
}

int b5_next_raw(b5_file_t *f, uint8_t **buf, uint64_t *len, uint64_t *cap, uint64_t *size) {
    return read_raw_here(f, buf, len, cap, size);
}

/* raw bytes of the record `read_id` (as b5_next_raw).  The record is not parsed here, so an index from disk is checked
 * (the size it records and the id stored in the record). */
int b5_get_raw(b5_file_t *f, const char *read_id, uint8_t **buf, uint64_t *len, uint64_t *cap, uint64_t *size) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!f->idx) {
            const int rc = index_build(f, attempt == 0);
            if (rc) return rc;
        }
        b5_idx_entry_t key;
        key.id = (char *)read_id;
        key.offset = 0;
        const b5_idx_entry_t *e = (const b5_idx_entry_t *)bsearch(&key, f->idx, f->n_idx, sizeof key, idx_cmp);
        if (!e) {
            if (f->idx_from_disk && attempt == 0) { idx_drop(f); continue; }
            return B5_ERR_NOTFOUND;
        }
        const uint64_t esize = e->size, len0 = *len;
        if (fseek(f->fp, (long)e->offset, SEEK_SET) != 0) return B5_ERR_IO;
        const int rc = read_raw_here(f, buf, len, cap, size);
        int fits = rc == 0 && *size + 8 == esize;
        if (fits && f->record_press == 0) {
            const uint8_t *r = *buf + len0;
            const size_t idl = strlen(read_id);
            uint16_t have;
            fits = *size >= 2 + idl;
            if (fits) {
                memcpy(&have, r, 2);
                fits = have == idl && memcmp(r + 2, read_id, idl) == 0;
            }
        } else if (fits) {
            /* compressed records: the id is inside the deflate stream.  An index of another file of that name can point
             * at ANOTHER record of the same compressed size; the record is inflated and parsed here to compare (read-id
             * mode fetches a handful of reads: the extra inflate does not matter) */
            uint8_t *sc = NULL;
            uint64_t sc_cap = 0;
            b5_view_t v;
            const int prc = b5_parse_raw(f, *buf + len0, *size, &sc, &sc_cap, &v);
            fits = prc == 0 && v.id_len == strlen(read_id) && memcmp(v.read_id, read_id, v.id_len) == 0;
            free(sc);
        }
        if (fits || !f->idx_from_disk || attempt == 1) return rc == 0 && !fits ? B5_ERR_FORMAT : rc;
        *len = len0;
        idx_drop(f);
    }
    return B5_ERR_FORMAT;
}

int b5_parse_raw(const b5_file_t *f, const uint8_t *raw, uint64_t size, uint8_t **scratch, uint64_t *scratch_cap,
                 b5_view_t *out) {
    const uint8_t *p = raw;
    uint64_t n = size;
    if (f->record_press == 1) {
        uint64_t want = *scratch_cap ? *scratch_cap : size * 4 + 256;  /* grown on Z_BUF_ERROR; kept small: one per record slot */
        for (;;) {
            const int rc = grow(scratch, scratch_cap, want);
            if (rc) return rc;
            uLongf outlen = *scratch_cap;
            const int z = uncompress(*scratch, &outlen, raw, size);
            if (z == Z_OK) { n = outlen; break; }
            if (z != Z_BUF_ERROR) return B5_ERR_PRESS;
            if (*scratch_cap >= B5_MAX_INFLATED) return B5_ERR_PRESS; /* zip bomb / truncated stream */
            want = *scratch_cap * 2;
        }
        p = *scratch;
    }
    if (n < 2) return B5_ERR_FORMAT;
    uint16_t idl;
    memcpy(&idl, p, 2);
    if (2 + (uint64_t)idl + 44 > n) return B5_ERR_FORMAT;
    out->rec = p;
    out->rec_len = n;
    out->read_id = (const char *)(p + 2);
    out->id_len = idl;
    const uint8_t *q = p + 2 + idl;
    memcpy(&out->read_group, q, 4);
    memcpy(&out->digitisation, q + 4, 8);
    memcpy(&out->offset, q + 12, 8);
    memcpy(&out->range, q + 20, 8);
    memcpy(&out->sampling_rate, q + 28, 8);
    uint64_t ln;
    memcpy(&ln, q + 36, 8);
    q += 44;
    const uint64_t left = n - (uint64_t)(q - p);
    out->signal = q;
    if (f->signal_press == 1) {
        if (ln > left || ln < 4 || ln > 0xffffffffull) return B5_ERR_FORMAT;
        uint32_t count;
        memcpy(&count, q, 4);
        if (count > 0x7fffffffu || 4 + ((uint64_t)count + 3) / 4 + (uint64_t)count > ln) return B5_ERR_FORMAT;
        out->signal_bytes = ln;
        out->n_samples = count;
    } else {
        if (ln > 0x7fffffffull || ln * 2 > left) return B5_ERR_FORMAT;
        out->signal_bytes = ln * 2;
        out->n_samples = (uint32_t)ln;
    }
    return 0;
}

int b5_parse_head(const b5_file_t *f, const uint8_t *raw, uint64_t size, uint8_t **scratch, uint64_t *scratch_cap,
                  b5_view_t *out) {
    if (f->record_press != 1) return B5_ERR_FORMAT;
    /* u16 id_len, id, 44 bytes of fields, and the blob's count word: 512 bytes hold an id of up to 462 characters;
     * a longer one (never seen: ids are UUIDs) goes round again with the room it needs */
    uint64_t want = *scratch_cap >= 512 ? *scratch_cap : 512;
    for (;;) {
        const int rc = grow(scratch, scratch_cap, want);
        if (rc) return rc;
        z_stream z;
        memset(&z, 0, sizeof z);
        if (inflateInit(&z) != Z_OK) return B5_ERR_MEM;
        z.next_in = (Bytef *)raw;
        z.avail_in = size > 0xffffffffull ? 0xffffffffu : (uInt)size;
        z.next_out = *scratch;
        z.avail_out = (uInt)(*scratch_cap > 0x7fffffffull ? 0x7fffffffull : *scratch_cap);
        const int zr = inflate(&z, Z_SYNC_FLUSH);
        const uint64_t got = z.total_out;
        inflateEnd(&z);
        if (zr != Z_OK && zr != Z_STREAM_END && zr != Z_BUF_ERROR) return B5_ERR_PRESS;
        if (got < 2) return B5_ERR_FORMAT;
        const uint8_t *p = *scratch;
        uint16_t idl;
        memcpy(&idl, p, 2);
        const uint64_t need = 2 + (uint64_t)idl + 44 + (f->signal_press == 1 ? 4 : 0);
        if (got < need) {
            if (zr == Z_STREAM_END) return B5_ERR_FORMAT;   /* the record ends inside its own head */
            if (*scratch_cap >= need) return B5_ERR_PRESS;    /* (room was not the problem) */
            want = need;
            continue;
        }
        out->rec = NULL;
        out->rec_len = 0;
        out->read_id = (const char *)(p + 2);
        out->id_len = idl;
        const uint8_t *q = p + 2 + idl;
        memcpy(&out->read_group, q, 4);
        memcpy(&out->digitisation, q + 4, 8);
        memcpy(&out->offset, q + 12, 8);
        memcpy(&out->range, q + 20, 8);
        memcpy(&out->sampling_rate, q + 28, 8);
        uint64_t ln;
        memcpy(&ln, q + 36, 8);
        out->signal = NULL;
        out->signal_offset = 2u + idl + 44u;
        if (f->signal_press == 1) {
            if (ln < 4 || ln > 0xffffffffull) return B5_ERR_FORMAT;
            uint32_t count;
            memcpy(&count, q + 44, 4);
            if (count > 0x7fffffffu || 4 + ((uint64_t)count + 3) / 4 + (uint64_t)count > ln) return B5_ERR_FORMAT;
            out->signal_bytes = ln;
            out->n_samples = count;
        } else {
            if (ln > 0x7fffffffull) return B5_ERR_FORMAT;
            out->signal_bytes = ln * 2;
            out->n_samples = (uint32_t)ln;
        }
        return 0;
    }
}

int64_t b5_aux_fixed_bytes(const b5_file_t *f) {
    /* the "#char*\tuint32_t\t..." line of the header names the type of every column: the first eight are the primary
     * fields, what follows are the auxiliary ones (slow5lib/src/slow5.c:794-881) */
    const char *p = f->hdr_text;
    const char *line = NULL;
    while (p && *p) {
        if (p[0] == '#' && strncmp(p, "#char*", 6) == 0) { line = p; break; }
        p = strchr(p, '\n');
        if (p) ++p;
    }
    if (!line) return -1;
    int64_t total = 0;
    int col = 0;
    const char *q = line + 1;
    for (;;) {
        const char *e = q;
        while (*e && *e != '\t' && *e != '\n') ++e;
        const size_t n = (size_t)(e - q);
        if (col >= 8) {
            int sz = -1;
            if (n && q[n - 1] == '*') sz = -1;                                   /* arrays: u64 length + data */
            else if ((n == 6 && !strncmp(q, "int8_t", 6)) || (n == 7 && !strncmp(q, "uint8_t", 7)) || (n == 4 && !strncmp(q, "char", 4))) sz = 1;
            else if ((n == 7 && !strncmp(q, "int16_t", 7)) || (n == 8 && !strncmp(q, "uint16_t", 8))) sz = 2;
            else if ((n == 7 && !strncmp(q, "int32_t", 7)) || (n == 8 && !strncmp(q, "uint32_t", 8)) || (n == 5 && !strncmp(q, "float", 5))) sz = 4;
            else if ((n == 7 && !strncmp(q, "int64_t", 7)) || (n == 8 && !strncmp(q, "uint64_t", 8)) || (n == 6 && !strncmp(q, "double", 6))) sz = 8;
            else if (n >= 5 && !strncmp(q, "enum{", 5) && q[n - 1] == '}') sz = 1;  /* enums are stored as uint8_t */
            if (sz < 0) return -1;
            total += sz;
        }
        ++col;
        if (*e != '\t') break;
        q = e + 1;
    }
    return col >= 8 ? total : -1;
}

int b5_svb_zd_decode(const uint8_t *blob, uint64_t nbytes, int16_t *dst, uint32_t count) {
    if (nbytes < 4) return B5_ERR_PRESS;
    uint32_t c;
    memcpy(&c, blob, 4);
    if (c != count) return B5_ERR_PRESS;
    const uint64_t nkeys = ((uint64_t)count + 3) / 4;
    if (4 + nkeys > nbytes) return B5_ERR_PRESS;
    const uint8_t *keys = blob + 4, *data = keys + nkeys, *end = blob + nbytes;
    int32_t prev = 0;
    for (uint32_t i = 0; i < count; i++) {
        const unsigned code = (keys[i >> 2] >> ((i & 3) * 2)) & 3u;
        if (data + code + 1 > end) return B5_ERR_PRESS;
        uint32_t v = 0;
        memcpy(&v, data, code + 1);
        data += code + 1;
        prev += (int32_t)(v >> 1) ^ -(int32_t)(v & 1);
        dst[i] = (int16_t)prev;
    }
    return data == end ? 0 : B5_ERR_PRESS;
}

/* ------------------------------------------------------------------ zero-copy sequential access */

int b5_map(b5_file_t *f) {
    if (f->map) return 0;
    struct stat st;
    if (fstat(fileno(f->fp), &st) != 0 || st.st_size <= 0) return B5_ERR_IO;
    void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(f->fp), 0);
    if (m == MAP_FAILED) return B5_ERR_IO;
    f->map = (const uint8_t *)m;
    f->map_len = (uint64_t)st.st_size;
    f->map_pos = f->first_rec;
    return 0;
}

int b5_next_ref(b5_file_t *f, const uint8_t **ptr, uint64_t *size) {
    if (!f->map) return B5_ERR_IO;
    const uint64_t left = f->map_len > f->map_pos ? f->map_len - f->map_pos : 0;
    const uint8_t *p = f->map + f->map_pos;
    if (left >= 5 && memcmp(p, B5_EOF_MARK, 5) == 0) return left == 5 ? B5_EOF : B5_ERR_FORMAT;
    if (left < 8) return B5_ERR_IO;
    uint64_t sz;
    memcpy(&sz, p, 8);
    if (sz == 0 || sz > (1ull << 36)) return B5_ERR_FORMAT;
    if (sz > left - 8) return B5_ERR_IO;
    *ptr = p + 8;
    *size = sz;
    f->map_pos += 8 + sz;
    return 0;
}
