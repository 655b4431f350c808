/* blow5.h -- minimal BLOW5 reader of the sigtk-amd host CLI.
 *
 * Written from the on-disk layout; the reference reads the same files
 * through slow5lib (slow5_open / slow5_get_next / slow5_get, slow5lib/include/slow5/slow5.h:345-454).
 * Supports record compression none/zlib and signal compression none/svb-zd; zstd files are
 * rejected (the reference build here has no zstd either).  Auxiliary fields are skipped. */
#ifndef SGK_BLOW5_H
#define SGK_BLOW5_H

#include <stdint.h>
#include <stdio.h>

typedef struct {
    char *read_id;
    uint32_t read_group;
    double digitisation, offset, range, sampling_rate;
    uint64_t len_raw_signal;
    int16_t *raw_signal; /* capacity grows; reused across b5_next calls like slow5_rec_t */
    uint64_t cap_signal;
    uint8_t *buf;        /* scratch for the (de)compressed record */
    uint64_t cap_buf;
    uint8_t *zbuf;
    uint64_t cap_zbuf;
} b5_rec_t;

typedef struct {
    char *id;
    uint64_t offset; /* file offset of the u64 record size */
    uint64_t size;   /* 8 + record bytes (what the reference's .idx stores) */
} b5_idx_entry_t;

typedef struct {
    FILE *fp;
    char *path;
    uint8_t version[3];
    uint8_t record_press; /* 0 none, 1 zlib */
    uint8_t signal_press; /* 0 none, 1 svb-zd */
    uint32_t num_read_groups;
    char *hdr_text;       /* header text block */
    uint32_t hdr_size;    /* its length; the file starts with 68 fixed bytes followed by it */
    uint64_t first_rec;   /* file offset of the first record */
    b5_idx_entry_t *idx;  /* built lazily by b5_index (sorted by id) */
    uint64_t n_idx;
    int idx_from_disk;    /* the index was loaded from "<file>.idx" (checked against what it points at) */
    const uint8_t *map;   /* the whole file mapped read-only (b5_map), or NULL */
    uint64_t map_len, map_pos;
} b5_file_t;

#define B5_EOF (-1)
#define B5_ERR_IO (-2)
#define B5_ERR_FORMAT (-3)
#define B5_ERR_PRESS (-4)
#define B5_ERR_MEM (-5)
#define B5_ERR_NOTFOUND (-6)

b5_file_t *b5_open(const char *path);
void b5_close(b5_file_t *f);
/* value of header attribute `name` for read group `rg`, or NULL (malloc'd; caller frees) */
char *b5_hdr_get(const b5_file_t *f, const char *name, uint32_t rg);
/* next record in file order: 0 ok, B5_EOF at the proper end, other negatives on error */
int b5_next(b5_file_t *f, b5_rec_t *rec);
/* the read-id index: loaded from "<file>.idx" (the reference's on-disk index, slow5lib/src/slow5_idx.c) when that
 * exists and matches the file, else built by one sequential scan and written there (best effort) */
int b5_index(b5_file_t *f);
/* random access by read id (needs b5_index) */
int b5_get(b5_file_t *f, const char *read_id, b5_rec_t *rec);
void b5_rec_free(b5_rec_t *rec);

/* ---- split API for the pipelined reader --------------------------------------
 * One thread pulls the records' bytes off the file in order (b5_next_raw / b5_get_raw); any number
 * of threads then inflate + parse them (b5_parse_raw, re-entrant) -- the split slow5lib's compiled-out
 * batch API makes between slow5_get_next_mem and slow5_rec_depress_parse (slow5_mt.c:252-317). */
typedef struct {
    const char *read_id;  /* NOT NUL-terminated: id_len bytes inside the parsed record */
    uint16_t id_len;
    uint32_t read_group;
    double digitisation, offset, range, sampling_rate;
    const uint8_t *rec;    /* the whole (inflated) record and its length: qts rewrites the signal and keeps the rest */
    uint64_t rec_len;
    const uint8_t *signal; /* svb-zd blob (signal_press 1) or little-endian int16 samples */
    uint64_t signal_bytes;
    uint32_t n_samples;    /* from the blob's count word, or signal_bytes / 2 */
    uint32_t signal_offset; /* of the signal inside the inflated record (b5_parse_head: `rec` / `signal` are NULL there) */
} b5_view_t;

/* appends the next record's on-disk bytes (without the u64 size) to *buf at *len (realloc'd as needed);
 * *size receives their length.  0 ok, B5_EOF, or an error. */
int b5_next_raw(b5_file_t *f, uint8_t **buf, uint64_t *len, uint64_t *cap, uint64_t *size);
int b5_get_raw(b5_file_t *f, const char *read_id, uint8_t **buf, uint64_t *len, uint64_t *cap, uint64_t *size);
/* Zero-copy variant of b5_next_raw: maps the file once (b5_map; returns non-zero if that is not possible, the
 * caller then stays with b5_next_raw) and hands out pointers into the mapping, valid until b5_close.  The pages
 * are first touched by whoever parses the record, i.e. by the thread pool rather than by the reader. */
int b5_map(b5_file_t *f);
int b5_next_ref(b5_file_t *f, const uint8_t **ptr, uint64_t *size);
/* inflates (if the file uses zlib records) into *scratch (realloc'd as needed) and parses; the view points
 * into *scratch or into raw.  Re-entrant: touches no state of f besides its compression settings. */
int b5_parse_raw(const b5_file_t *f, const uint8_t *raw, uint64_t size, uint8_t **scratch, uint64_t *scratch_cap,
                 b5_view_t *out);
/* The HEAD of a zlib-compressed record only: inflates just far enough (into *scratch) for the id, the scaling, the signal's
 * byte length and -- svb-zd -- its sample count; the view's read_id points into *scratch, rec / signal are NULL,
 * signal_offset tells where the signal starts in the inflated record.  A few microseconds per record instead of the
 * whole inflate: the record itself is inflated on the GPU (sgk_job_begin_zrec).  Files with zlib records only. */
int b5_parse_head(const b5_file_t *f, const uint8_t *raw, uint64_t size, uint8_t **scratch, uint64_t *scratch_cap,
                  b5_view_t *out);
/* bytes of a record's auxiliary fields when all of them are primitive (fixed size), else -1 (an array field, an unknown
 * type): with a fixed size the inflated length of a record follows from its head */
int64_t b5_aux_fixed_bytes(const b5_file_t *f);
/* scalar streamvbyte + zigzag-delta decode of one blob into dst[count] (host-decode fallback path) */
int b5_svb_zd_decode(const uint8_t *blob, uint64_t nbytes, int16_t *dst, uint32_t count);

#endif
