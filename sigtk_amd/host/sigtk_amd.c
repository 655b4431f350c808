/* sigtk_amd.c -- `sigtk-amd`: drop-in host CLI for sigtk's per-record subtools
 * (pa / event / stat / jnn / prefix) with the compute on MI355X through libsigtk_gpu.so.
 *
 * Mirrors the observable behaviour of the reference front-end:
 *   src/main.c:76-123   subcommand dispatch, version/usage, stderr footer
 *   src/cmain.c:40-156  options (-h -V -n -c --print-stat, ignored -o/--verbose), DNA/RNA and pore
 *                       detection from read-group 0, sequential or read-id mode
 *   src/cfunc.c         the TSV grammar of every subtool (byte-identical output)
 * What differs by design: records are not processed one at a time.  A reader thread pulls the records'
 * bytes off the file in order and a pool of threads inflates/parses them into the pinned staging of a
 * job (sgk_job_*, include/sigtk_gpu.h); svb-zd signals are handed to the GPU still compressed and
 * decoded there.  Several batches are in flight (one per GPU plus two), a writer thread formats the
 * rows of finished batches on a pool of threads (exact fast number formatting, fmt.h) and emits them
 * in file order.  Whole batches are the multi-GPU sharding unit: no collective.
 * Extra options: --gpus N, --batch-samples M, -t/--threads T, --host-decode.
 */
#include <getopt.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include <unistd.h>
#include <zlib.h>

#include "blow5.h"
#include "fmt.h"
#include "sigtk_gpu.h"

#define SIGTK_VERSION "0.2.0" /* the reference version whose CLI this mirrors (src/sigtk.h:11) */

#define INFO(fn, msg) fprintf(stderr, "[%s::INFO]\033[1;34m %s\033[0m\n", fn, msg)
#define WARNING(fn, ...)                                         \
    do {                                                         \
        fprintf(stderr, "[%s::WARNING]\033[1;33m ", fn);         \
        fprintf(stderr, __VA_ARGS__);                            \
        fprintf(stderr, "\033[0m\n");                            \
    } while (0)
#define ERROR(fn, ...)                                           \
    do {                                                         \
        fprintf(stderr, "[%s::ERROR]\033[1;31m ", fn);           \
        fprintf(stderr, __VA_ARGS__);                            \
        fprintf(stderr, "\033[0m\n");                            \
    } while (0)

typedef struct {
    int8_t rna, compact, p_stat, pore; /* opt_t, src/sigtk.h:115-120 */
} opt_t;

enum { MODE_EVENT, MODE_STAT, MODE_PREFIX, MODE_JNN, MODE_PA, MODE_ENT, MODE_QTS };

static double realtime(void) {
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return tp.tv_sec + tp.tv_usec * 1e-6;
}
/* SGK_CLI_TIMING: where the wall time of a run goes that the pipeline's stage sums do not show (process start,
 * HIP initialisation, first launch = code object load, buffer set-up, teardown).  Seconds since the process was
 * started (execve), from /proc: starttime of /proc/self/stat against /proc/uptime. */
static double since_exec(void) {
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[1024];
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');   /* the command name may hold spaces */
    if (!p) return -1.0;
    unsigned long long start = 0;
    int field = 2;
    for (p++; *p && field < 22; p++)
        if (*p == ' ') field++;
    if (sscanf(p, "%llu", &start) != 1) return -1.0;
    double up = 0.0;
    f = fopen("/proc/uptime", "r");
    if (!f) return -1.0;
    if (fscanf(f, "%lf", &up) != 1) up = -1.0;
    fclose(f);
    return up < 0 ? -1.0 : up - (double)start / (double)sysconf(_SC_CLK_TCK);
}
static double g_t_main = 0.0, g_exec_to_main = -1.0, g_t_first_submit = 0.0, g_t_pipeline_end = 0.0, g_t_destroyed = 0.0;

static double cputime(void) {
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
}
static long peakrss(void) {
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    return r.ru_maxrss * 1024;
}

/* ------------------------------------------------------------------ header-derived options */

/* src/misc.c:34-60 */
static int8_t drna_detect(const b5_file_t *f) {
    char *exp = b5_hdr_get(f, "experiment_type", 0);
    int8_t rna = 0;
    if (!exp) {
        WARNING("drna_detect", "%s", "experiment_type not found in SLOW5 header. Assuming genomic_dna");
        return 0;
    }
    if (strcmp(exp, "genomic_dna") == 0) {
        INFO("drna_detect", "DNA data detected.");
    } else if (strcmp(exp, "rna") == 0) {
        rna = 1;
        INFO("drna_detect", "RNA data detected.");
    } else {
        WARNING("drna_detect", "Unknown experiment type: %s. Assuming genomic_dna", exp);
    }
    for (uint32_t i = 1; i < f->num_read_groups; i++) {
        char *cur = b5_hdr_get(f, "experiment_type", i);
        if (cur && strcmp(cur, exp))
            WARNING("drna_detect", "Experiment type mismatch: %s != %s in read group %d. Defaulted to %s", cur, exp,
                    (int)i, exp);
        free(cur);
    }
    free(exp);
    return rna;
}

/* src/misc.c:74-101 */
static int8_t pore_detect(const b5_file_t *f) {
    char *kit = b5_hdr_get(f, "sequencing_kit", 0);
    int8_t pore = SGK_PORE_R9;
    if (!kit) {
        WARNING("pore_detect", "%s", "sequencing_kit not found in SLOW5 header. Assuming R9.4.1");
        return 0;
    }
    if (strstr(kit, "114")) {
        pore = SGK_PORE_R10;
        INFO("pore_detect", "R10 data detected.");
    } else if (strstr(kit, "rna004")) {
        pore = SGK_PORE_RNA004;
        INFO("pore_detect", "RNA004 data detected.");
    } else {
        INFO("pore_detect", "R9 data detected.");
    }
    for (uint32_t i = 1; i < f->num_read_groups; i++) {
        char *cur = b5_hdr_get(f, "sequencing_kit", i);
        if (cur && strcmp(cur, kit))
            WARNING("pore_detect", "sequencing_kit type mismatch: %s != %s in read group %d. Defaulted to %s", cur,
                    kit, (int)i, kit);
        free(cur);
    }
    free(kit);
    return pore;
}

/* ------------------------------------------------------------------ small utilities */

/* Fatal errors can be raised on the reader, loader and writer threads while other threads still have HIP work in
 * flight: leave without running exit handlers (exit() would run the HIP runtime's concurrently with live streams). */
static void die_now(void) {
    fflush(stderr);
    _exit(EXIT_FAILURE);
}

static void die_mem(void) {
    ERROR("main", "%s", "out of memory");
    die_now();
}

/* growable output buffer */
typedef struct {
    char *p;
    size_t n, cap;
} sbuf_t;

static inline char *sbuf_room(sbuf_t *b, size_t need) {
    if (b->n + need > b->cap) {
        size_t c = b->cap ? b->cap : (1u << 16);
        while (c < b->n + need) c *= 2;
        b->p = (char *)realloc(b->p, c);
        if (!b->p) die_mem();
        b->cap = c;
    }
    return b->p + b->n;
}
static inline void sbuf_str(sbuf_t *b, const char *s, size_t len) {
    memcpy(sbuf_room(b, len), s, len);
    b->n += len;
}

/* parallel for with dynamic scheduling on a persistent pool: fn(ctx, i, tid) for i in [0, n).  A pool belongs to
 * one caller thread (the loader and the writer each own one); its workers sleep between calls. */
typedef void (*pfor_fn)(void *ctx, uint32_t i, int tid);
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    pthread_t *th;
    int nworkers;        /* threads besides the caller */
    uint64_t epoch;      /* bumped for every pfor call */
    int running;         /* workers still inside the current call */
    int quit;
    pfor_fn fn;
    void *ctx;
    uint32_t n, next;
} pool_t;
typedef struct {
    pool_t *pool;
    int tid;
} pool_worker_t;

static void pool_drain(pool_t *p, int tid) {
    for (;;) {
        const uint32_t i = __atomic_fetch_add(&p->next, 1u, __ATOMIC_RELAXED);
        if (i >= p->n) break;
        p->fn(p->ctx, i, tid);
    }
}
static void *pool_worker(void *a_) {
    pool_worker_t *w = (pool_worker_t *)a_;
    pool_t *p = w->pool;
    uint64_t seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->quit && p->epoch == seen) pthread_cond_wait(&p->go, &p->mu);
        if (p->quit) break;
        seen = p->epoch;
        pthread_mutex_unlock(&p->mu);
        pool_drain(p, w->tid);
        pthread_mutex_lock(&p->mu);
        if (--p->running == 0) pthread_cond_signal(&p->done);
    }
    pthread_mutex_unlock(&p->mu);
    free(w);
    return NULL;
}
static pool_t *pool_create(int nthreads) {
    pool_t *p = (pool_t *)calloc(1, sizeof *p);
    if (!p) die_mem();
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->go, NULL);
    pthread_cond_init(&p->done, NULL);
    p->nworkers = nthreads > 1 ? nthreads - 1 : 0;
    p->th = (pthread_t *)calloc((size_t)p->nworkers + 1, sizeof(pthread_t));
    if (!p->th) die_mem();
    for (int t = 0; t < p->nworkers; t++) {
        pool_worker_t *w = (pool_worker_t *)malloc(sizeof *w);
        if (!w) die_mem();
        w->pool = p;
        w->tid = t + 1;
        if (pthread_create(&p->th[t], NULL, pool_worker, w) != 0) {
            ERROR("pool", "%s", "cannot create thread");
            die_now();
        }
    }
    return p;
}
static void pool_destroy(pool_t *p) {
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->nworkers; t++) pthread_join(p->th[t], NULL);
    free(p->th);
    free(p);
}
static void pfor(pool_t *p, uint32_t n, pfor_fn fn, void *ctx) {
    if (n == 0) return;
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->ctx = ctx; p->n = n; p->next = 0;
    p->running = p->nworkers;
    p->epoch++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    pool_drain(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->running > 0) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
}

/* ------------------------------------------------------------------ batches and the pipeline
 *
 *   reader thread                 loader (main thread)                       writer thread
 *   -------------                 --------------------                       -------------
 *   take a free batch             take the next filled batch                 take the oldest submitted batch
 *   read the records' bytes       N threads: inflate + parse                 sgk_job_wait
 *   off the file, in order        sgk_job_begin (layout, pinned staging)     N threads: format rows into chunks
 *   hand it over                  N threads: copy svb-zd blobs / samples     fwrite the chunks in order
 *                                 sgk_job_submit (async H2D, decode,         return the batch to the free list
 *                                 kernels, D2H)
 *
 * n_gpus + 3 batches circulate; batch k runs on GPU k mod n_gpus (whole batches are the sharding unit:
 * every output row depends on one record only, src/cmain.c:118-120, so there is no collective). */

typedef struct {
    uint64_t raw_off, raw_size; /* the record's on-disk bytes inside batch_t.raw ... */
    const uint8_t *raw_ptr;     /* ... or, when the file is mapped, in the mapping */
    uint8_t *scratch;           /* inflated record (kept per slot; grows only) */
    uint64_t scratch_cap;
    b5_view_t v;
    int err;
} lrec_t;

typedef struct batch {
    sgk_job_t *job;
    sgk_job_input_t in;
    lrec_t *recs;
    uint32_t n, cap;
    uint8_t *raw;
    uint64_t raw_len, raw_cap;
    uint64_t bytes;  /* on-disk bytes gathered so far (the batch limit applies to it) */
    uint32_t *lengths, *blob_bytes;
    uint32_t *sig_off, *sig_len, *room; /* zrec: where the signal sits in the inflated record, its bytes, the record's inflated size */
    int svb;        /* signal staged as svb-zd blobs (GPU decode) */
    int zrec;       /* whole zlib records staged: inflated and decoded on the GPU */
    int last;       /* sentinel: no more batches */
    struct batch *next;
} batch_t;

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    batch_t *head, *tail;
} queue_t;

static void q_init(queue_t *q) {
    pthread_mutex_init(&q->mu, NULL);
    pthread_cond_init(&q->cv, NULL);
    q->head = q->tail = NULL;
}
static void q_push(queue_t *q, batch_t *b) {
    pthread_mutex_lock(&q->mu);
    b->next = NULL;
    if (q->tail) q->tail->next = b;
    else q->head = b;
    q->tail = b;
    pthread_cond_signal(&q->cv);
    pthread_mutex_unlock(&q->mu);
}
static batch_t *q_pop(queue_t *q) {
    pthread_mutex_lock(&q->mu);
    while (!q->head) pthread_cond_wait(&q->cv, &q->mu);
    batch_t *b = q->head;
    q->head = b->next;
    if (!q->head) q->tail = NULL;
    pthread_mutex_unlock(&q->mu);
    return b;
}

static batch_t *q_try_pop(queue_t *q) {
    pthread_mutex_lock(&q->mu);
    batch_t *b = q->head;
    if (b) {
        q->head = b->next;
        if (!q->head) q->tail = NULL;
    }
    pthread_mutex_unlock(&q->mu);
    return b;
}

typedef struct {
    b5_file_t *f;
    int mode, nthreads, host_decode;
    int zrec;            /* records go to the GPU as they sit in the file (zlib records, svb-zd signal, fixed-size auxiliary fields) */
    int64_t aux_bytes;   /* ... the bytes of a record's auxiliary fields then */
    opt_t opt;
    queue_t free_q, filled_q, ready_q;
    /* reader thread input */
    uint64_t limit_bytes;
    char **ids;      /* read-id mode: ids[0..n_ids) */
    int n_ids;
    /* qts */
    int q_bits, q_method;
    FILE *out_fp;    /* rows / records go here (stdout except for qts) */
    pool_t *load_pool; /* the loader's worker pool (the writer creates its own) */
    struct batch *pool; /* all batch slots; [0, n_created) have a job */
    int n_created, n_max, n_gpus;
    uint64_t batches_read;
    double t_read, t_parse, t_stage, t_wait, t_format, t_write; /* --verbose timing */
    uint64_t n_long_declined; /* long reads the 16-workgroup path of stat / jnn / prefix declined (redone on one wavefront) */
    uint64_t n_reads, n_samples;
} pipe_t;

static void gpu_fail(const char *what, int rc) {
    ERROR(what, "%s %s", sgk_strerror(rc), sgk_last_hip_error());
    die_now();
}

static void batch_add_record(batch_t *b, uint64_t size, const uint8_t *ref) {
    if (b->n == b->cap) {
        const uint32_t nc = b->cap ? b->cap * 2 : 1024;
        b->recs = (lrec_t *)realloc(b->recs, sizeof(lrec_t) * nc);
        b->lengths = (uint32_t *)realloc(b->lengths, sizeof(uint32_t) * nc);
        b->blob_bytes = (uint32_t *)realloc(b->blob_bytes, sizeof(uint32_t) * nc);
        b->sig_off = (uint32_t *)realloc(b->sig_off, sizeof(uint32_t) * nc);
        b->sig_len = (uint32_t *)realloc(b->sig_len, sizeof(uint32_t) * nc);
        b->room = (uint32_t *)realloc(b->room, sizeof(uint32_t) * nc);
        if (!b->recs || !b->lengths || !b->blob_bytes || !b->sig_off || !b->sig_len || !b->room) die_mem();
        memset(b->recs + b->cap, 0, sizeof(lrec_t) * (nc - b->cap));
        b->cap = nc;
    }
    b->recs[b->n].raw_off = b->raw_len - size;
    b->recs[b->n].raw_size = size;
    b->recs[b->n].raw_ptr = ref;
    b->bytes += size;
    b->n++;
}

/* phase 1 (parallel): inflate + parse record i */
typedef struct {
    pipe_t *P;
    batch_t *b;
} lctx_t;
static void load_parse(void *ctx_, uint32_t i, int tid) {
    (void)tid;
    lctx_t *c = (lctx_t *)ctx_;
    lrec_t *r = &c->b->recs[i];
    const uint8_t *raw = r->raw_ptr ? r->raw_ptr : c->b->raw + r->raw_off;
    /* zrec: the head only (id, scaling, sizes: a few hundred inflated bytes); the record itself is inflated on the GPU */
    if (c->P->zrec) r->err = b5_parse_head(c->P->f, raw, r->raw_size, &r->scratch, &r->scratch_cap, &r->v);
    else r->err = b5_parse_raw(c->P->f, raw, r->raw_size, &r->scratch, &r->scratch_cap, &r->v);
    if (c->P->zrec && !r->err && (r->raw_size > 0xffffffffull || r->v.signal_offset + r->v.signal_bytes + (uint64_t)c->P->aux_bytes > 0xffffffffull))
        r->err = B5_ERR_FORMAT;
}
/* phase 2 (parallel): stage record i's signal and scaling into the job's pinned buffers */
static void load_stage(void *ctx_, uint32_t i, int tid) {
    (void)tid;
    lctx_t *c = (lctx_t *)ctx_;
    batch_t *b = c->b;
    lrec_t *r = &b->recs[i];
    b->in.digitisation[i] = r->v.digitisation;
    b->in.offset[i] = r->v.offset;
    b->in.range[i] = r->v.range;
    if (b->zrec) {
        memcpy(b->in.blobs + b->in.blob_offsets[i], r->raw_ptr ? r->raw_ptr : b->raw + r->raw_off, r->raw_size);
    } else if (b->svb) {
        memcpy(b->in.blobs + b->in.blob_offsets[i], r->v.signal, r->v.signal_bytes);
    } else if (c->P->f->signal_press == 1) {
        r->err = b5_svb_zd_decode(r->v.signal, r->v.signal_bytes, b->in.samples + b->in.offsets[i], r->v.n_samples);
    } else {
        memcpy(b->in.samples + b->in.offsets[i], r->v.signal, r->v.signal_bytes);
    }
}

/* parse + stage + submit the records gathered in b */
static void batch_launch(pipe_t *P, batch_t *b) {
    lctx_t c = {P, b};
    double t0 = realtime();
    pfor(P->load_pool, b->n, load_parse, &c);
    for (uint32_t i = 0; i < b->n; i++) {
        if (b->recs[i].err) {
            fprintf(stderr, "Error in slow5_get_next. Error code %d\n", b->recs[i].err);
            die_now();
        }
        b->lengths[i] = b->recs[i].v.n_samples;
        b->blob_bytes[i] = P->zrec ? (uint32_t)b->recs[i].raw_size : (uint32_t)b->recs[i].v.signal_bytes;
        if (P->zrec) {
            b->sig_off[i] = b->recs[i].v.signal_offset;
            b->sig_len[i] = (uint32_t)b->recs[i].v.signal_bytes;
            b->room[i] = (uint32_t)(b->recs[i].v.signal_offset + b->recs[i].v.signal_bytes + (uint64_t)P->aux_bytes);
        }
        P->n_samples += b->recs[i].v.n_samples;
    }
    P->n_reads += b->n;
    double t1 = realtime();
    P->t_parse += t1 - t0;
    b->svb = P->f->signal_press == 1 && !P->host_decode;
    b->zrec = P->zrec;
    int rc;
    if (b->zrec) rc = sgk_job_begin_zrec(b->job, b->n, b->lengths, b->blob_bytes, b->sig_off, b->sig_len, b->room, &b->in);
    else rc = sgk_job_begin(b->job, b->n, b->lengths, b->svb ? SGK_SIGNAL_SVBZD : SGK_SIGNAL_INT16, b->blob_bytes, &b->in);
    if (rc != SGK_OK) gpu_fail("sgk_job_begin", rc);
    pfor(P->load_pool, b->n, load_stage, &c);
    for (uint32_t i = 0; i < b->n; i++) {
        if (b->recs[i].err) {
            fprintf(stderr, "Error in slow5_get_next. Error code %d\n", b->recs[i].err);
            die_now();
        }
    }
    int tool = SGK_TOOL_PA, flags = 0;
    switch (P->mode) {
        case MODE_EVENT: tool = SGK_TOOL_EVENT; flags = P->opt.compact ? SGK_JOB_EVENTS_LENGTHS : 0; break;
        case MODE_STAT: tool = SGK_TOOL_STAT; break;
        case MODE_PREFIX: tool = SGK_TOOL_PREFIX; break;
        case MODE_JNN: tool = SGK_TOOL_JNN; break;
        case MODE_ENT: tool = SGK_TOOL_ENT; break;
        default: break;
    }
    if (P->mode == MODE_QTS)
        rc = sgk_job_submit_qts(b->job, P->q_bits, P->q_method, P->f->signal_press == 1 ? SGK_SIGNAL_SVBZD : SGK_SIGNAL_INT16);
    else
        rc = sgk_job_submit(b->job, tool, P->opt.rna, P->opt.pore, flags);
    if (rc != SGK_OK) gpu_fail("sgk_job_submit", rc);
    P->t_stage += realtime() - t1;
    if (g_t_first_submit == 0.0) g_t_first_submit = realtime();
    q_push(&P->ready_q, b);
}

/* ------------------------------------------------------------------ row formatters (src/cfunc.c)
 * Byte-for-byte the reference's printf output; numbers go through fmt.h. */

#define ID_OF(b, r) (b)->recs[r].v.read_id, (b)->recs[r].v.id_len

static inline void put_id_len(sbuf_t *o, const batch_t *b, uint32_t r) {
    char *p = sbuf_room(o, (size_t)b->recs[r].v.id_len + 32);
    memcpy(p, b->recs[r].v.read_id, b->recs[r].v.id_len);
    p += b->recs[r].v.id_len;
    *p++ = '\t';
    p = fmt_u64(p, b->lengths[r]);
    *p++ = '\t';
    o->n = (size_t)(p - o->p);
}

/* print_events, cfunc.c:16-61 */
static void row_events(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r, opt_t opt) {
    const uint64_t a = out->slots[r], n = out->counts[r];
    const uint32_t *ln = out->ev_length + a;
    if (opt.compact) {
        /* (the job hands back the lengths alone, SGK_JOB_EVENTS_LENGTHS: the events of a read are contiguous from
         * sample 0, events.c:491-501, so event[0].start is 0 and the last event's end the sum of the lengths) */
        put_id_len(o, b, r);
        if (n) {
            uint64_t end = 0;
            for (uint64_t j = 0; j < n; j++) end += ln[j];
            char *p = sbuf_room(o, 96 + n * 12);
            p = fmt_u64(p, 0); *p++ = '\t';
            p = fmt_u64(p, end); *p++ = '\t';
            p = fmt_u64(p, n); *p++ = '\t';
            for (uint64_t j = 0; j < n; j++) {
                const int mi = (int)ln[j];
                if (mi) {
                    p = fmt_i64(p, mi);
                    if (j < n - 1) *p++ = ',';
                }
            }
            o->n = (size_t)(p - o->p);
        } else {
            sbuf_str(o, ".\t.\t.\t.", 7);
        }
        sbuf_str(o, "\n", 1);
    } else {
        const uint32_t *st = out->ev_start + a;
        const float *mean = out->ev_mean + a, *sd = out->ev_stdv + a;
        const size_t idl = b->recs[r].v.id_len;
        char *p = sbuf_room(o, n * (idl + 160) + 8);
        for (uint64_t j = 0; j < n; j++) {
            memcpy(p, b->recs[r].v.read_id, idl);
            p += idl;
            *p++ = '\t';
            p = fmt_i64(p, (int)j); *p++ = '\t';
            p = fmt_u64(p, st[j]); *p++ = '\t';
            p = fmt_u64(p, (uint64_t)st[j] + ln[j]); *p++ = '\t';
            p = fmt_f6(p, mean[j]); *p++ = '\t';
            p = fmt_f6(p, sd[j]); *p++ = '\n';
        }
        *p++ = '\n'; /* cfunc.c:58 */
        o->n = (size_t)(p - o->p);
    }
}

/* jnn_print, jnn.c:309-350 */
static void row_jnn(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r, opt_t opt) {
    put_id_len(o, b, r);
    const uint64_t len = b->lengths[r];
    if (len > 0) {
        const uint64_t a = out->slots[r], n = out->counts[r];
        const int32_t *x = out->seg_x + a, *y = out->seg_y + a;
        char *p = sbuf_room(o, 32 + n * 48);
        p = fmt_i64(p, (int)n); *p++ = '\t';
        if (opt.compact) {
            uint64_t ci = 0, mi = 0;
            for (uint64_t i = 0; i < n; i++) {
                ci += (mi = (uint64_t)x[i] - ci);
                if (mi) { p = fmt_i64(p, (int)mi); *p++ = 'H'; }
                ci += (mi = (uint64_t)y[i] - ci);
                if (mi) { p = fmt_i64(p, (int)mi); *p++ = ','; }
            }
        } else {
            for (uint64_t i = 0; i < n; i++) {
                p = fmt_i64(p, x[i]); *p++ = ',';
                p = fmt_i64(p, y[i]); *p++ = ';';
            }
        }
        if (n == 0) *p++ = '.';
        o->n = (size_t)(p - o->p);
    }
    sbuf_str(o, "\n", 1);
}

/* stat_func, cfunc.c:126-159 */
static void row_stat(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r) {
    put_id_len(o, b, r);
    const sgk_stat_rec_t *s = &out->stat[r];
    char *p = sbuf_room(o, 6 * 52);
    p = fmt_f6(p, s->raw_mean); *p++ = '\t';
    p = fmt_f6(p, s->pa_mean); *p++ = '\t';
    p = fmt_f6(p, s->raw_std); *p++ = '\t';
    p = fmt_f6(p, s->pa_std); *p++ = '\t';
    p = fmt_i64(p, (int)(int16_t)s->raw_median); *p++ = '\t';
    p = fmt_f6(p, s->pa_median); *p++ = '\t';
    *p++ = '\n';
    o->n = (size_t)(p - o->p);
}

/* prefix_func, cfunc.c:169-234 */
static void row_prefix(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r, opt_t opt) {
    put_id_len(o, b, r);
    const sgk_prefix_rec_t *q = &out->prefix[r];
    char *p = sbuf_room(o, 512);
    if (q->adapt_y > 0) {
        p = fmt_i64(p, q->adapt_x); *p++ = '\t';
        p = fmt_i64(p, q->adapt_y); *p++ = '\t';
        if (q->polya_y > 0) {
            p = fmt_i64(p, (int64_t)q->polya_x + q->adapt_y); *p++ = '\t';
            p = fmt_i64(p, (int64_t)q->polya_y + q->adapt_y);
        } else {
            memcpy(p, ".\t.", 3); p += 3;
        }
        if (opt.p_stat) {
            *p++ = '\t';
            p = fmt_f6(p, q->adapt_mean); *p++ = '\t';
            p = fmt_f6(p, q->adapt_std); *p++ = '\t';
            p = fmt_f6(p, q->adapt_median); *p++ = '\t';
            if (q->polya_y > 0) {
                *p++ = '\t';
                p = fmt_f6(p, q->polya_mean); *p++ = '\t';
                p = fmt_f6(p, q->polya_std); *p++ = '\t';
                p = fmt_f6(p, q->polya_median); *p++ = '\t';
            } else {
                memcpy(p, "\t.\t.\t.", 6); p += 6;
            }
        }
    } else {
        memcpy(p, ".\t.\t.\t.", 7); p += 7;
    }
    *p++ = '\n';
    o->n = (size_t)(p - o->p);
}

/* pa_func, cfunc.c:85-102 */
static void row_pa(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r) {
    put_id_len(o, b, r);
    const uint64_t len = b->lengths[r];
    const float *pa = out->pa + out->offsets[r];
    char *p = sbuf_room(o, len * 50 + 8);
    for (uint64_t i = 0; i < len; i++) {
        p = fmt_f6(p, pa[i]);
        if (i != len - 1) *p++ = ',';
    }
    *p++ = '\n';
    o->n = (size_t)(p - o->p);
}

/* entmain's row, ent.c:108-163: id, then "%f" of three doubles (the histograms come from the GPU, the sum over
 * the non-empty bins is sgk_ent_finish: the reference's own arithmetic in the reference's order) */
static void row_ent(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r) {
    double e[3];
    const uint64_t off = out->offsets[r];
    sgk_ent_finish(&out->ent[r], out->ent_over_raw ? out->ent_over_raw + off : NULL,
                   out->ent_over_delta ? out->ent_over_delta + off : NULL, e);
    char *p = sbuf_room(o, (size_t)b->recs[r].v.id_len + 3 * 330 + 8);
    memcpy(p, b->recs[r].v.read_id, b->recs[r].v.id_len);
    p += b->recs[r].v.id_len;
    p += sprintf(p, "\t%f\t%f\t%f\n", e[0], e[1], e[2]);
    o->n = (size_t)(p - o->p);
}

/* qts (src/qts.c:118-150): the record as it was read, with the signal replaced by the quantised one (and
 * len_raw_signal with it); everything else -- id, read group, scaling, auxiliary fields -- is kept byte for byte.
 * Emitted as the file stores it: u64 size, then the record (one zlib stream when the file compresses records). */
static __thread uint8_t *qts_tmp;
static __thread size_t qts_tmp_cap;
static void row_qts(sbuf_t *o, const batch_t *b, const sgk_job_output_t *out, uint32_t r, const b5_file_t *f) {
    const b5_view_t *v = &b->recs[r].v;
    const size_t head = (size_t)(v->signal - v->rec) - 8; /* up to the u64 len_raw_signal */
    const uint8_t *tail = v->signal + v->signal_bytes;
    const size_t tail_len = (size_t)(v->rec + v->rec_len - tail);
    const uint8_t *sig;
    uint64_t sig_bytes, len_field;
    if (f->signal_press == 1) {
        sig = out->qts_blobs + out->qts_blob_offsets[r];
        sig_bytes = out->qts_blob_lengths[r];
        len_field = sig_bytes;
    } else {
        sig = (const uint8_t *)(out->qts_samples + out->offsets[r]);
        sig_bytes = (uint64_t)b->lengths[r] * 2;
        len_field = b->lengths[r];
    }
    const size_t n = head + 8 + (size_t)sig_bytes + tail_len;
    if (f->record_press == 1) {
        if (qts_tmp_cap < n) {
            qts_tmp = (uint8_t *)realloc(qts_tmp, n + n / 4 + 64);
            if (!qts_tmp) die_mem();
            qts_tmp_cap = n + n / 4 + 64;
        }
        uint8_t *t = qts_tmp;
        memcpy(t, v->rec, head);
        memcpy(t + head, &len_field, 8);
        memcpy(t + head + 8, sig, (size_t)sig_bytes);
        memcpy(t + head + 8 + sig_bytes, tail, tail_len);
        uLongf zlen = compressBound((uLong)n);
        char *p = sbuf_room(o, 8 + (size_t)zlen);
        if (compress2((Bytef *)p + 8, &zlen, t, (uLong)n, Z_DEFAULT_COMPRESSION) != Z_OK) {
            ERROR("qts", "%s", "zlib compression failed");
            die_now();
        }
        const uint64_t z64 = zlen;
        memcpy(p, &z64, 8);
        o->n += 8 + (size_t)zlen;
    } else {
        char *p = sbuf_room(o, 8 + n);
        const uint64_t n64 = n;
        memcpy(p, &n64, 8);
        memcpy(p + 8, v->rec, head);
        memcpy(p + 8 + head, &len_field, 8);
        memcpy(p + 16 + head, sig, (size_t)sig_bytes);
        memcpy(p + 16 + head + sig_bytes, tail, tail_len);
        o->n += 8 + n;
    }
}

/* ------------------------------------------------------------------ writer */

typedef struct {
    pipe_t *P;
    batch_t *b;
    sgk_job_output_t out;
    sbuf_t *chunk;          /* one buffer per chunk of reads */
    uint32_t *chunk_lo;     /* n_chunks + 1 */
} wctx_t;

static void write_chunk(void *ctx_, uint32_t k, int tid) {
    (void)tid;
    wctx_t *c = (wctx_t *)ctx_;
    sbuf_t *o = &c->chunk[k];
    o->n = 0;
    const opt_t opt = c->P->opt;
    for (uint32_t r = c->chunk_lo[k]; r < c->chunk_lo[k + 1]; r++) {
        switch (c->P->mode) {
            case MODE_EVENT: row_events(o, c->b, &c->out, r, opt); break;
            case MODE_JNN: row_jnn(o, c->b, &c->out, r, opt); break;
            case MODE_STAT: row_stat(o, c->b, &c->out, r); break;
            case MODE_PREFIX: row_prefix(o, c->b, &c->out, r, opt); break;
            case MODE_ENT: row_ent(o, c->b, &c->out, r); break;
            case MODE_QTS: row_qts(o, c->b, &c->out, r, c->P->f); break;
            default: row_pa(o, c->b, &c->out, r); break;
        }
    }
}

static void *writer_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    sbuf_t *chunk = NULL;
    uint32_t *chunk_lo = NULL;
    uint32_t chunk_cap = 0;
    const int wthreads = P->nthreads > 32 ? 32 : P->nthreads;  /* formatting is light: fewer threads */
    pool_t *wpool = pool_create(wthreads);
    for (;;) {
        batch_t *b = q_pop(&P->ready_q);
        if (b->last) break;
        double t0 = realtime();
        int rc = sgk_job_wait(b->job);
        if (rc == SGK_ERR_FORMAT && b->zrec) {
            /* a record that does not inflate (or not to what its head announced), or a blob that does not decode: what the
             * reference reports for it (slow5_get_next's negative return, src/cmain.c:121-124) */
            fprintf(stderr, "Error in slow5_get_next. Error code %d\n", B5_ERR_PRESS);
            die_now();
        }
        if (rc != SGK_OK) gpu_fail("sgk_job_wait", rc);
        P->n_long_declined += sgk_job_long_declined(b->job);
        wctx_t c;
        c.P = P;
        c.b = b;
        rc = sgk_job_output(b->job, &c.out);
        if (rc != SGK_OK) gpu_fail("sgk_job_output", rc);
        double t1 = realtime();
        P->t_wait += t1 - t0;
        /* contiguous chunks of reads with about equal sample counts; a few per thread for balance */
        uint32_t nchunks = (uint32_t)wthreads * 4;
        if (nchunks > b->n) nchunks = b->n;
        if (nchunks > chunk_cap) {
            chunk = (sbuf_t *)realloc(chunk, sizeof(sbuf_t) * nchunks);
            chunk_lo = (uint32_t *)realloc(chunk_lo, sizeof(uint32_t) * ((size_t)nchunks + 1));
            if (!chunk || !chunk_lo) die_mem();
            memset(chunk + chunk_cap, 0, sizeof(sbuf_t) * (nchunks - chunk_cap));
            chunk_cap = nchunks;
        }
        uint64_t total = 0;
        for (uint32_t r = 0; r < b->n; r++) total += b->lengths[r] + 64;
        uint64_t acc = 0;
        uint32_t k = 0;
        chunk_lo[0] = 0;
        for (uint32_t r = 0; r < b->n && k + 1 < nchunks; r++) {
            acc += b->lengths[r] + 64;
            if (acc * nchunks >= total * (uint64_t)(k + 1)) chunk_lo[++k] = r + 1;
        }
        while (k < nchunks) chunk_lo[++k] = b->n;
        c.chunk = chunk;
        c.chunk_lo = chunk_lo;
        pfor(wpool, nchunks, write_chunk, &c);
        double t2 = realtime();
        P->t_format += t2 - t1;
        for (uint32_t i = 0; i < nchunks; i++)
            if (chunk[i].n && fwrite(chunk[i].p, 1, chunk[i].n, P->out_fp) != chunk[i].n) {
                ERROR("writer", "%s", "write to the output failed");
                die_now();
            }
        P->t_write += realtime() - t2;
        b->n = 0;
        b->raw_len = 0;
        b->bytes = 0;
        q_push(&P->free_q, b);
    }
    pool_destroy(wpool);
    for (uint32_t i = 0; i < chunk_cap; i++) free(chunk[i].p);
    free(chunk);
    free(chunk_lo);
    return NULL;
}

/* ------------------------------------------------------------------ reader thread
 * Pulls the records' bytes off the file (sequentially, or by read id) into free batches and hands each full batch
 * to the loader, so that file reads overlap with inflating / staging the previous batch. */
/* the reader's next empty batch: a free one, else a new one if the input has earned it, else wait */
static batch_t *take_free_batch(pipe_t *P) {
    batch_t *b = q_try_pop(&P->free_q);
    if (!b && P->n_created < P->n_max && P->batches_read >= 4u * (uint64_t)P->n_created) {
        b = &P->pool[P->n_created];
        const int rc = sgk_job_create(P->n_created % P->n_gpus, &b->job);
        if (rc != SGK_OK) gpu_fail("sgk_job_create", rc);
        P->n_created++;
    }
    if (!b) b = q_pop(&P->free_q);
    P->batches_read++;
    return b;
}

static void *reader_main(void *arg) {
    pipe_t *P = (pipe_t *)arg;
    b5_file_t *f = P->f;
    batch_t *b = take_free_batch(P);
    int ret = 0;
    if (P->n_ids == 0) {
        const int mapped = b5_map(f) == 0;  /* zero-copy: the pool touches the pages when it inflates the records */
        for (;;) {
            uint64_t size = 0;
            const uint8_t *ref = NULL;
            const double t0 = realtime();
            ret = mapped ? b5_next_ref(f, &ref, &size) : b5_next_raw(f, &b->raw, &b->raw_len, &b->raw_cap, &size);
            P->t_read += realtime() - t0;
            if (ret < 0) break;
            batch_add_record(b, size, ref);
            if (b->bytes >= P->limit_bytes) {
                q_push(&P->filled_q, b);
                b = take_free_batch(P);
            }
        }
        if (ret != B5_EOF) {
            fprintf(stderr, "Error in slow5_get_next. Error code %d\n", ret);
            die_now();
        }
    } else {
        if (b5_index(f) < 0) {
            ERROR("cmain", "Error loading index file for %s", f->path);
            die_now();
        }
        for (int i = 0; i < P->n_ids; i++) {
            fprintf(stderr, "Read ID %s\n", P->ids[i]);
            uint64_t size = 0;
            if (b5_get_raw(f, P->ids[i], &b->raw, &b->raw_len, &b->raw_cap, &size) < 0) {
                ERROR("cmain", "%s", "Error when fetching the read");
                die_now();
            }
            batch_add_record(b, size, NULL);
            if (b->bytes >= P->limit_bytes) {
                q_push(&P->filled_q, b);
                b = take_free_batch(P);
            }
        }
    }
    b->last = 1;  /* the final (possibly empty) batch ends the stream */
    q_push(&P->filled_q, b);
    return NULL;
}

/* ------------------------------------------------------------------ the pipeline driver */
static void run_pipeline(pipe_t *P, int n_gpus, double t_init) {
    q_init(&P->free_q);
    q_init(&P->filled_q);
    q_init(&P->ready_q);
    /* Up to n_gpus + 3 batches (one being read, one being inflated/staged, n_gpus in flight, one being written),
     * but only two to begin with: a job's pinned and device buffers cost tens of milliseconds to set up, which a
     * small input never earns back.  The reader adds a batch when it would otherwise wait and the input has
     * already run to several batches per existing one (take_free_batch). */
    const int nbatch = n_gpus + 3;
    batch_t *pool = (batch_t *)calloc((size_t)nbatch + 1, sizeof(batch_t));
    if (!pool) die_mem();
    const double t_jobs0 = realtime();
    P->pool = pool;
    P->n_max = nbatch;
    P->n_gpus = n_gpus;
    P->n_created = nbatch < 2 ? nbatch : 2;
    for (int i = 0; i < P->n_created; i++) {
        const int rc = sgk_job_create(i % n_gpus, &pool[i].job);
        if (rc != SGK_OK) gpu_fail("sgk_job_create", rc);
        q_push(&P->free_q, &pool[i]);
    }
    const double t_jobs = realtime() - t_jobs0;
    P->load_pool = pool_create(P->nthreads);
    pthread_t wth, rth;
    if (pthread_create(&wth, NULL, writer_main, P) != 0 || pthread_create(&rth, NULL, reader_main, P) != 0) {
        ERROR("cmain", "%s", "cannot create the pipeline threads");
        die_now();
    }
    /* loader: inflate/parse, stage and submit every filled batch */
    for (;;) {
        batch_t *b = q_pop(&P->filled_q);
        const int last = b->last;
        b->last = 0;
        if (b->n) batch_launch(P, b);
        else q_push(&P->free_q, b);
        if (last) break;
    }
    pthread_join(rth, NULL);
    pool[nbatch].last = 1;
    q_push(&P->ready_q, &pool[nbatch]);
    pthread_join(wth, NULL);
    pool_destroy(P->load_pool);
    g_t_pipeline_end = realtime();
    if (P->n_long_declined)
        WARNING("pipeline", "%lu long read(s) were declined by the long-read path (a barrier wait timed out) and redone on one "
                "wavefront each; the output is unaffected", (unsigned long)P->n_long_declined);
    if (getenv("SGK_CLI_TIMING"))
        fprintf(stderr,
                "[sigtk-amd] %lu reads, %lu samples, %d threads, %d GPU(s): read %.3f s, inflate+parse %.3f s, "
                "stage+submit %.3f s | wait-for-GPU %.3f s, format %.3f s, write %.3f s | HIP init %.3f s, job create %.3f s\n",
                (unsigned long)P->n_reads, (unsigned long)P->n_samples, P->nthreads, n_gpus, P->t_read, P->t_parse,
                P->t_stage, P->t_wait, P->t_format, P->t_write, t_init, t_jobs);
    /* The process is about to leave through _exit (main): the jobs' pinned and device buffers go with it.  Releasing
     * them one hipHostFree / hipFree at a time costs more than a small input's whole pipeline (SGK_CLI_TIMING shows
     * it), so they are only released when asked to (leak checkers: SGK_CLI_FREE=1). */
    if (getenv("SGK_CLI_FREE")) {
        for (int i = 0; i < P->n_created; i++) {
            sgk_job_destroy(pool[i].job);
            for (uint32_t k = 0; k < pool[i].cap; k++) free(pool[i].recs[k].scratch);
            free(pool[i].recs); free(pool[i].raw); free(pool[i].lengths); free(pool[i].blob_bytes); free(pool[i].sig_off); free(pool[i].sig_len); free(pool[i].room);
        }
        free(pool);
    }
    g_t_destroyed = realtime();
}

/* ------------------------------------------------------------------ cmain (src/cmain.c:40-156) */

/* host threads worth starting: half the online CPUs, but no more than the affinity mask or a cgroup CPU quota allow
 * (a GPU box may show 256 CPUs and grant a job 16: 64 threads then only throttle each other -- 1e10 samples of `stat`
 * took 3.7 s with 64 threads and 3.1 s with 16, profiles/r05_k_cli_steady_before.json) */
static int host_thread_budget(void) {
    long nc = sysconf(_SC_NPROCESSORS_ONLN);
    int n = nc > 1 ? (int)(nc / 2) : 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int a = CPU_COUNT(&set);
        if (a > 0 && a < n) n = a;
    }
    FILE *fp = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (fp) {
        char q[64];
        long period = 0;
        if (fscanf(fp, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long quota = atol(q);
            const int c = (int)((quota + period - 1) / period);
            if (c > 0 && c < n) n = c;
        }
        fclose(fp);
    }
    if (n > 64) n = 64;  /* inflate scales to ~64 threads; beyond that thread start-up dominates */
    return n < 1 ? 1 : n;
}

static struct option long_options[] = {
    {"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},       {"version", no_argument, 0, 'V'},
    {"output", required_argument, 0, 'o'},  {"print-stat", no_argument, 0, 0},   {"no-header", no_argument, 0, 'n'},
    {"compact", no_argument, 0, 'c'},       {"gpus", required_argument, 0, 0},   {"batch-samples", required_argument, 0, 0},
    {"threads", required_argument, 0, 't'}, {"host-decode", no_argument, 0, 0},  {"host-inflate", no_argument, 0, 0},
    {0, 0, 0, 0}};

static int cmain(int argc, char *argv[], const char *mode_s) {
    /* `ent` has its own front end in the reference (src/ent.c:63-105): only -h/-V (and --no-header) are options,
     * exactly one file argument, no DNA/RNA or pore detection */
    const int is_ent = strcmp(mode_s, "ent") == 0;
    const char *optstring = is_ent ? "hVt:" : "o:hVnct:";
    int longindex = 0, c;
    FILE *fp_help = stderr;
    int8_t hdr = 1;
    opt_t opt = {0, 0, 0, 0};
    int n_gpus = 1, nthreads = 0, host_decode = 0, host_inflate = 0, batch_set = 0;
    /* default batch: small (16 M samples) where the GPU stage is short -- a job's buffers are then cheap to set up
     * and the host stages overlap sooner; 64 M for jnn / prefix, whose one-read-per-lane kernels take as long for
     * a small batch as for a large one */
    uint64_t batch_samples = (strcmp(mode_s, "jnn") == 0 || strcmp(mode_s, "prefix") == 0) ? 64ull << 20 : 16ull << 20;

    while ((c = getopt_long(argc, argv, optstring, long_options, &longindex)) >= 0) {
        if (c == 'V') {
            fprintf(stdout, "sigtk %s\n", SIGTK_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'n') {
            hdr = 0;
        } else if (c == 'c') {
            opt.compact = 1;
        } else if (c == 't') {
            nthreads = atoi(optarg);
        } else if (c == 0 && longindex == 4) {
            opt.p_stat = 1;
        } else if (c == 0 && longindex == 7) {
            n_gpus = atoi(optarg);
        } else if (c == 0 && longindex == 8) {
            batch_samples = strtoull(optarg, NULL, 10);
            batch_set = 1;
        } else if (c == 0 && longindex == 10) {
            host_decode = 1;
        } else if (c == 0 && longindex == 11) {
            host_inflate = 1;
        }
    }
    if (is_ent && (argc - optind != 1 || fp_help == stdout)) {
        fprintf(fp_help, "Usage: sigtk ent a.blow5\n");
        fprintf(fp_help, "\nbasic options:\n");
        fprintf(fp_help, "   -h                         help\n");
        fprintf(fp_help, "   -n                         suppress header\n");
        fprintf(fp_help, "   --version                  print version\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    if (argc - optind < 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: sigtk %s reads.blow5 read_id1 read_id2 .. \n", mode_s);
        fprintf(fp_help, "       sigtk %s reads.blow5\n", mode_s);
        fprintf(fp_help, "\nbasic options:\n");
        fprintf(fp_help, "   -h                         help\n");
        fprintf(fp_help, "   -n                         suppress header\n");
        fprintf(fp_help, "   -c                         compact output\n");
        fprintf(fp_help, "   --version                  print version\n");
        fprintf(fp_help, "   --gpus INT                 number of GPUs; whole batches go round-robin [1]\n");
        fprintf(fp_help, "   --batch-samples INT        approximate raw samples per batch [134217728; with --host-inflate 16777216, jnn / prefix 67108864]\n");
        fprintf(fp_help, "   -t, --threads INT          host threads for inflating records / formatting rows [auto]\n");
        fprintf(fp_help, "   --host-decode              decode svb-zd signals on the host instead of the GPU\n");
        fprintf(fp_help, "   --host-inflate             inflate zlib records on the host threads instead of the GPU\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }

    b5_file_t *f = b5_open(argv[optind]);
    if (!f) {
        if (is_ent) fprintf(stderr, "Error in opening file\n"); /* ent.c:97-100 */
        else ERROR("cmain", "cannot open %s. ", argv[optind]);
        die_now();
    }
    if (!is_ent) {
        opt.rna = drna_detect(f);
        opt.pore = pore_detect(f);
    }

    int mode;
    if (strcmp(mode_s, "event") == 0) {
        mode = MODE_EVENT;
        if (hdr) {
            if (opt.compact) printf("read_id\tlen_raw_signal\traw_start\traw_end\tnum_event\tevents\n");
            else printf("read_id\tevent_idx\traw_start\traw_end\tevent_mean\tevent_std\n");
        }
    } else if (strcmp(mode_s, "stat") == 0) {
        mode = MODE_STAT;
        if (hdr) printf("read_id\tlen_raw_signal\traw_mean\tpa_mean\traw_std\tpa_std\traw_median\tpa_median\n");
    } else if (strcmp(mode_s, "prefix") == 0) {
        mode = MODE_PREFIX;
        if (hdr) {
            printf("read_id\tlen_raw_signal\tadapt_start\tadapt_end\tpolya_start\tpolya_end");
            if (opt.p_stat) printf("\tadapt_mean\tadapt_std\tadapt_median\tpolya_mean\tpolya_std\tpolya_median");
            printf("\n");
        }
    } else if (strcmp(mode_s, "jnn") == 0) {
        mode = MODE_JNN;
        if (hdr) printf("read_id\tlen_raw_signal\tnum_seg\tseg\n");
    } else if (is_ent) {
        mode = MODE_ENT;
        if (hdr) printf("read_id\traw_ent\tdelta_ent\tbyte_ent\n");
    } else {
        mode = MODE_PA;
        if (hdr) printf("read_id\tlen_raw_signal\tpa\n");
    }

    const double t_init0 = realtime();
    const int ndev = sgk_device_count();
    const double t_init = realtime() - t_init0;
    if (ndev <= 0) {
        ERROR("cmain", "%s", "no usable GPU: sigtk-amd has no CPU compute path");
        die_now();
    }
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > ndev) {
        WARNING("cmain", "--gpus %d requested but %d visible; using %d", n_gpus, ndev, ndev);
        n_gpus = ndev;
    }
    if (nthreads <= 0) nthreads = host_thread_budget();

    pipe_t P;
    memset(&P, 0, sizeof P);
    P.f = f;
    P.mode = mode;
    P.nthreads = nthreads;
    P.host_decode = host_decode;
    /* Records go to the GPU as they sit in the file when they are zlib streams around an svb-zd signal and their
     * auxiliary fields have a fixed size (the inflated length of a record then follows from its head); everything else
     * -- and --host-inflate / --host-decode -- is inflated by the host threads as before. */
    P.aux_bytes = b5_aux_fixed_bytes(f);
    P.zrec = !host_inflate && !host_decode && f->record_press == 1 && f->signal_press == 1 && P.aux_bytes >= 0;
    /* ... in batches of 128 M samples: the inflate kernel is a wavefront per record and takes ~35 ms however many records
     * it has (up to the ~4 800 the GPU holds at once), so a batch should bring a thousand of them (1e10 samples of `stat`:
     * 5.2 s with the 16 M-sample batches of the host path, 1.7 s with 128 M, profiles/r05_cli_steady.json) */
    if (P.zrec && !batch_set) batch_samples = 128ull << 20;
    P.opt = opt;
    P.out_fp = stdout;
    P.limit_bytes = batch_samples;
    P.ids = argv + optind + 1;
    P.n_ids = argc - optind - 1;
    run_pipeline(&P, n_gpus, t_init);
    fflush(stdout);
    b5_close(f);
    return 0;
}

/* ------------------------------------------------------------------ qtsmain (src/qts.c:46-165)
 * Same options as the reference (-o FILE, -b INT in [1,8], -m floor|round|fill-ones).  The output keeps the
 * input's header block and compression settings verbatim (the reference re-serialises the header through slow5lib
 * and always writes zlib + svb-zd); records carry the same fields, auxiliary data included, with the quantised
 * signal -- quantised and re-encoded (svb-zd) on the GPU, deflated on the host thread pool. */
static struct option qts_long_options[] = {
    {"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},   {"version", no_argument, 0, 'V'},
    {"output", required_argument, 0, 'o'},  {"bits", required_argument, 0, 'b'}, {"method", required_argument, 0, 'm'},
    {"gpus", required_argument, 0, 0},      {"batch-samples", required_argument, 0, 0}, {"threads", required_argument, 0, 't'},
    {0, 0, 0, 0}};

static int qtsmain(int argc, char *argv[]) {
    const char *optstring = "hVv:o:b:m:t:";
    int longindex = 0, c;
    FILE *fp_help = stderr;
    char *out_fn = NULL;
    int b = 1, n_gpus = 1, nthreads = 0;
    const char *method = "round";
    uint64_t batch_samples = 16ull << 20; /* small batches: the buffers of a job are cheap to set up, the host stages overlap sooner */
    while ((c = getopt_long(argc, argv, optstring, qts_long_options, &longindex)) >= 0) {
        if (c == 'V') {
            fprintf(stdout, "sigtk %s\n", SIGTK_VERSION);
            exit(EXIT_SUCCESS);
        } else if (c == 'h') {
            fp_help = stdout;
        } else if (c == 'o') {
            out_fn = optarg;
        } else if (c == 'b') {
            b = atoi(optarg);
            if (b < 1 || b > 8) {
                fprintf(stderr, "Error: number of bits to truncate must be between 1 and 8\n");
                die_now();
            }
        } else if (c == 'm') {
            method = optarg;
        } else if (c == 't') {
            nthreads = atoi(optarg);
        } else if (c == 0 && longindex == 6) {
            n_gpus = atoi(optarg);
        } else if (c == 0 && longindex == 7) {
            batch_samples = strtoull(optarg, NULL, 10);
        }
    }
    if (argc - optind != 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: sigtk qts a.blow5 -o out.blow5\n");
        fprintf(fp_help, "\nbasic options:\n");
        fprintf(fp_help, "   -h                            help\n");
        fprintf(fp_help, "   -o FILE                       output file\n");
        fprintf(fp_help, "   --version                     print version\n");
        fprintf(fp_help, "   -b INT                        number of lower significant bits to eliminate [%d]\n", b);
        fprintf(fp_help, "   -m [floor|round|fill-ones]    quantisation method [round]\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }
    if (out_fn == NULL) {
        fprintf(stderr, "Error: output file not specified\n");
        die_now();
    }
    int q_method;
    if (strcmp(method, "floor") == 0) q_method = SGK_QTS_FLOOR;
    else if (strcmp(method, "round") == 0) q_method = SGK_QTS_ROUND;
    else if (strcmp(method, "fill-ones") == 0) q_method = SGK_QTS_FILL_ONES;
    else {
        fprintf(stderr, "Unknown method for -m. Available options are floor,round,fill-ones.\n");
        die_now();
    }
    b5_file_t *f = b5_open(argv[optind]);
    if (!f) {
        fprintf(stderr, "Error in opening file\n");
        die_now();
    }
    FILE *out = fopen(out_fn, "wb");
    if (!out) {
        fprintf(stderr, "Error opening file!\n");
        die_now();
    }
    /* header block: the fixed 68 bytes and the header text, as they are */
    {
        const size_t hb = 68 + (size_t)f->hdr_size;
        uint8_t *h = (uint8_t *)malloc(hb);
        if (!h) die_mem();
        if (fseek(f->fp, 0, SEEK_SET) != 0 || fread(h, 1, hb, f->fp) != hb || fwrite(h, 1, hb, out) != hb ||
            fseek(f->fp, (long)f->first_rec, SEEK_SET) != 0) {
            fprintf(stderr, "Error writing header!\n");
            die_now();
        }
        free(h);
    }
    const double t_init0 = realtime();
    const int ndev = sgk_device_count();
    const double t_init = realtime() - t_init0;
    if (ndev <= 0) {
        ERROR("qtsmain", "%s", "no usable GPU: sigtk-amd has no CPU compute path");
        die_now();
    }
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > ndev) n_gpus = ndev;
    if (nthreads <= 0) nthreads = host_thread_budget();
    pipe_t P;
    memset(&P, 0, sizeof P);
    P.f = f;
    P.mode = MODE_QTS;
    P.nthreads = nthreads;
    P.q_bits = b;
    P.q_method = q_method;
    P.out_fp = out;
    P.limit_bytes = batch_samples;
    run_pipeline(&P, n_gpus, t_init);
    if (fwrite("5WOLB", 1, 5, out) != 5 || fclose(out) != 0) {
        fprintf(stderr, "Error writing record!\n");
        die_now();
    }
    b5_close(f);
    return 0;
}

/* hidden helper for tests: dump id, length, scaling and a checksum of every record.  `--split` goes
 * through the pipelined reader's split API (b5_next_raw + b5_parse_raw + b5_svb_zd_decode; with --id: b5_get_raw). */
static uint64_t fnv_i16(const int16_t *x, uint64_t n) {
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; i++) {
        h ^= (uint16_t)x[i];
        h *= 1099511628211ull;
    }
    return h;
}
static int dumpmain(int argc, char *argv[]) {
    int split = 0;
    const char *path = NULL, *want_id = NULL;
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--split") == 0) split = 1;
        else if (strcmp(argv[i], "--map") == 0) split = 2;  /* split API over the mapped file (b5_map / b5_next_ref) */
        else if (strcmp(argv[i], "--id") == 0 && i + 1 < argc) want_id = argv[++i]; /* one record through the index */
        else path = argv[i];
    }
    if (!path) return 1;
    b5_file_t *f = b5_open(path);
    if (!f) {
        ERROR("dumpmain", "cannot open %s. ", path);
        return 1;
    }
    int ret;
    if (want_id && split) {
        /* the pipeline's read-id path: b5_get_raw (index + verification of the fetched record) + b5_parse_raw */
        uint8_t *raw = NULL, *scratch = NULL;
        uint64_t len = 0, cap = 0, scap = 0, size = 0;
        ret = b5_get_raw(f, want_id, &raw, &len, &cap, &size);
        b5_view_t v;
        if (ret == 0) ret = b5_parse_raw(f, raw, size, &scratch, &scap, &v);
        if (ret == 0) {
            int16_t *sig = (int16_t *)malloc(sizeof(int16_t) * (v.n_samples ? v.n_samples : 1));
            if (!sig) ret = -1;
            else if (f->signal_press == 1) ret = b5_svb_zd_decode(v.signal, v.signal_bytes, sig, v.n_samples);
            else if (v.signal_bytes != 2 * (uint64_t)v.n_samples) ret = -1;  /* (a malformed record: the copy would overrun) */
            else memcpy(sig, v.signal, v.signal_bytes);
            if (ret == 0)
                printf("%.*s\t%lu\t%.17g\t%.17g\t%.17g\t%016lx\n", (int)v.id_len, v.read_id, (unsigned long)v.n_samples,
                       v.digitisation, v.offset, v.range, (unsigned long)fnv_i16(sig, v.n_samples));
            free(sig);
        }
        free(raw);
        free(scratch);
        b5_close(f);
        return ret == 0 ? 0 : 1;
    }
    if (want_id) {
        b5_rec_t rec;
        memset(&rec, 0, sizeof rec);
        ret = b5_get(f, want_id, &rec);
        if (ret == 0)
            printf("%s\t%lu\t%.17g\t%.17g\t%.17g\t%016lx\n", rec.read_id, (unsigned long)rec.len_raw_signal,
                   rec.digitisation, rec.offset, rec.range, (unsigned long)fnv_i16(rec.raw_signal, rec.len_raw_signal));
        b5_rec_free(&rec);
        b5_close(f);
        return ret == 0 ? 0 : 1;
    }
    printf("#press\t%d\t%d\tgroups\t%u\n", f->record_press, f->signal_press, f->num_read_groups);
    if (split) {
        uint8_t *raw = NULL, *scratch = NULL;
        uint64_t raw_len = 0, raw_cap = 0, scratch_cap = 0, size = 0, sig_cap = 0;
        int16_t *sig = NULL;
        if (split == 2 && b5_map(f) != 0) {
            ERROR("dumpmain", "%s", "cannot map the file");
            return 1;
        }
        const uint8_t *ref = NULL;
        while ((ret = split == 2 ? b5_next_ref(f, &ref, &size) : b5_next_raw(f, &raw, &raw_len, &raw_cap, &size)) >= 0) {
            b5_view_t v;
            ret = b5_parse_raw(f, split == 2 ? ref : raw + raw_len - size, size, &scratch, &scratch_cap, &v);
            if (ret < 0) break;
            raw_len = 0;
            if (v.n_samples > sig_cap) {
                sig = (int16_t *)realloc(sig, sizeof(int16_t) * ((size_t)v.n_samples + 1));
                if (!sig) die_mem();
                sig_cap = v.n_samples;
            }
            if (f->signal_press == 1) {
                ret = b5_svb_zd_decode(v.signal, v.signal_bytes, sig, v.n_samples);
                if (ret < 0) break;
            } else if (v.n_samples) {
                memcpy(sig, v.signal, v.signal_bytes);
            }
            printf("%.*s\t%lu\t%.17g\t%.17g\t%.17g\t%016lx\n", (int)v.id_len, v.read_id, (unsigned long)v.n_samples,
                   v.digitisation, v.offset, v.range, (unsigned long)fnv_i16(sig, v.n_samples));
        }
        free(raw); free(scratch); free(sig);
    } else {
        b5_rec_t rec;
        memset(&rec, 0, sizeof rec);
        while ((ret = b5_next(f, &rec)) >= 0)
            printf("%s\t%lu\t%.17g\t%.17g\t%.17g\t%016lx\n", rec.read_id, (unsigned long)rec.len_raw_signal,
                   rec.digitisation, rec.offset, rec.range, (unsigned long)fnv_i16(rec.raw_signal, rec.len_raw_signal));
        b5_rec_free(&rec);
    }
    b5_close(f);
    return ret == B5_EOF ? 0 : 1;
}

/* hidden helper for tests: fmt_f6 / fmt_i64 against snprintf on every stride-th float bit pattern */
static int fmtcheckmain(int argc, char *argv[]) {
    /* _fmtcheck [stride [first last]]: every stride-th float bit pattern in [first, last] (defaults: all) */
    const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], NULL, 10) : 9973u;
    const uint64_t first = argc > 3 ? strtoull(argv[2], NULL, 0) : 0, last = argc > 3 ? strtoull(argv[3], NULL, 0) : 0xffffffffull;
    uint64_t bad = 0, n = 0;
    char a[512], b[512];
    for (uint64_t u = first; u <= last; u += stride ? stride : 1) {
        const uint32_t w = (uint32_t)u;
        float f;
        memcpy(&f, &w, 4);
        *fmt_f6(a, f) = '\0';
        snprintf(b, sizeof b, "%f", f);
        n++;
        if (strcmp(a, b) != 0 && bad++ < 10) printf("MISMATCH %08x: %s vs %s\n", w, a, b);
    }
    /* ties and carries: k / 2^m around the 6th decimal, values just below integers */
    for (int m = 1; m <= 30; m++)
        for (int k = 1; k < 4000; k += 2) {
            const float f = (float)k / (float)(1u << m);
            const float g[4] = {f, -f, f + 123456.0f, 999999.0f + f};
            for (int t = 0; t < 4; t++) {
                *fmt_f6(a, g[t]) = '\0';
                snprintf(b, sizeof b, "%f", g[t]);
                n++;
                if (strcmp(a, b) != 0 && bad++ < 10) printf("MISMATCH %a: %s vs %s\n", g[t], a, b);
            }
        }
    const int64_t iv[] = {0, 1, -1, 9, 10, 99, 100, 12345, -98765, 2147483647LL, -2147483648LL, 4294967295LL,
                          9223372036854775807LL, (-9223372036854775807LL - 1)};
    for (size_t i = 0; i < sizeof iv / sizeof iv[0]; i++) {
        *fmt_i64(a, iv[i]) = '\0';
        snprintf(b, sizeof b, "%ld", (long)iv[i]);
        n++;
        if (strcmp(a, b) != 0 && bad++ < 10) printf("MISMATCH int: %s vs %s\n", a, b);
    }
    printf("checked %lu values, %lu mismatches\n", (unsigned long)n, (unsigned long)bad);
    return bad ? 1 : 0;
}

/* ------------------------------------------------------------------ main (src/main.c:49-123) */

static void print_usage(FILE *fp) {
    fprintf(fp, "Usage: sigtk <command> [options]\n\n");
    fprintf(fp, "command:\n");
    fprintf(fp, "         pa        print raw signal in pico-amperes\n");
    fprintf(fp, "         event     segment raw signal into events\n");
    fprintf(fp, "         stat      print statistics of the raw signal\n");
    fprintf(fp, "         prefix    prefix segments such as adaptor and polyA\n");
    fprintf(fp, "         jnn       print segments found using JNN segmenter\n");
    fprintf(fp, "         ent       calculate entropies\n");
    fprintf(fp, "         qts       quantise the raw signal in a S/BLOW5 files\n");
    fprintf(fp, "\n(sigtk-amd: the per-read raw-signal subtools on MI355X; sref/ss are not part of it)\n");
    exit(fp == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
}

int main(int argc, char *argv[]) {
    const double realtime0 = realtime();
    g_t_main = realtime0;
    if (getenv("SGK_CLI_TIMING")) g_exec_to_main = since_exec();
    int ret = 1;
    if (argc < 2) {
        print_usage(stderr);
    } else if (strcmp(argv[1], "event") == 0 || strcmp(argv[1], "stat") == 0 || strcmp(argv[1], "prefix") == 0 ||
               strcmp(argv[1], "pa") == 0 || strcmp(argv[1], "jnn") == 0 || strcmp(argv[1], "ent") == 0) {
        ret = cmain(argc - 1, argv + 1, argv[1]);
    } else if (strcmp(argv[1], "qts") == 0) {
        ret = qtsmain(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "_dump") == 0) {
        return dumpmain(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "_fmtcheck") == 0) {
        return fmtcheckmain(argc - 1, argv + 1);
    } else if (strcmp(argv[1], "--version") == 0 || strcmp(argv[1], "-V") == 0) {
        fprintf(stdout, "sigtk %s\n", SIGTK_VERSION);
        exit(EXIT_SUCCESS);
    } else if (strcmp(argv[1], "--help") == 0 || strcmp(argv[1], "-h") == 0) {
        print_usage(stdout);
    } else {
        fprintf(stderr, "[sigtk] Unrecognised command %s\n", argv[1]);
        print_usage(stderr);
    }
    fprintf(stderr, "[%s] Version: %s\n", __func__, SIGTK_VERSION);
    fprintf(stderr, "[%s] CMD:", __func__);
    for (int i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
    fprintf(stderr, "\n[%s] Real time: %.3f sec; CPU time: %.3f sec; Peak RAM: %.3f GB\n\n", __func__,
            realtime() - realtime0, cputime(), peakrss() / 1024.0 / 1024.0 / 1024.0);
    if (getenv("SGK_CLI_TIMING") && g_t_pipeline_end > 0.0)
        fprintf(stderr,
                "[sigtk-amd] timeline: exec -> main %.3f s (loader, library constructors) | main -> first batch submitted "
                "%.3f s (HIP init, job buffers, first launches = code object load) | -> pipeline drained %.3f s | teardown "
                "%.3f s | total since exec %.3f s\n",
                g_exec_to_main, g_t_first_submit > 0.0 ? g_t_first_submit - g_t_main : -1.0,
                g_t_pipeline_end - (g_t_first_submit > 0.0 ? g_t_first_submit : g_t_main), g_t_destroyed - g_t_pipeline_end,
                g_exec_to_main + (realtime() - g_t_main));
    /* everything is written; leave without running the HIP runtime's exit handlers (they take longer than a
     * small input does) */
    if (fflush(stdout) != 0 || ferror(stdout)) {
        /* a short final write (ENOSPC, EPIPE) must not leave a truncated TSV behind exit code 0 */
        fprintf(stderr, "[%s::ERROR] writing to stdout failed\n", __func__);
        ret = EXIT_FAILURE;
    }
    fflush(stderr);
    _exit(ret);
}
