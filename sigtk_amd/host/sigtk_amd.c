/* sigtk_amd.c -- `sigtk-amd`: drop-in host CLI for sigtk's per-record subtools
 * (pa / event / stat / jnn / prefix) with the compute on MI355X through libsigtk_gpu.so.
 *
 * Mirrors the observable behaviour of the reference front-end:
 *   src/main.c:76-123   subcommand dispatch, version/usage, stderr footer
 *   src/cmain.c:40-156  options (-h -V -n -c --print-stat, ignored -o/--verbose), DNA/RNA and pore
 *                       detection from read-group 0, sequential or read-id mode
 *   src/cfunc.c         the TSV grammar of every subtool (byte-identical output)
 * What differs by design: records are not processed one at a time.  The reader fills a batch
 * (structure of arrays), the batch is split across the selected GPUs by cumulative sample count
 * (one host thread per GPU, no collective), and rows are printed in file order.
 * Extra options: --gpus N (default 1), --batch-samples M (default 64M samples per batch).
 */
#include <getopt.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <sys/time.h>

#include "blow5.h"
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

enum { MODE_EVENT, MODE_STAT, MODE_PREFIX, MODE_JNN, MODE_PA };

static double realtime(void) {
    struct timeval tp;
    gettimeofday(&tp, NULL);
    return tp.tv_sec + tp.tv_usec * 1e-6;
}
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

/* ------------------------------------------------------------------ batch */

typedef struct {
    char **ids;
    uint64_t *offsets; /* n+1 (CSR into samples) */
    double *dig, *off, *rng;
    int16_t *samples;
    uint32_t n, cap_reads;
    uint64_t cap_samples;
} batch_t;

static void die_mem(void) {
    ERROR("main", "%s", "out of memory");
    exit(EXIT_FAILURE);
}

static void batch_push(batch_t *b, const b5_rec_t *rec) {
    if (b->n == b->cap_reads) {
        b->cap_reads = b->cap_reads ? b->cap_reads * 2 : 1024;
        b->ids = (char **)realloc(b->ids, sizeof(char *) * b->cap_reads);
        b->offsets = (uint64_t *)realloc(b->offsets, sizeof(uint64_t) * ((size_t)b->cap_reads + 1));
        b->dig = (double *)realloc(b->dig, sizeof(double) * b->cap_reads);
        b->off = (double *)realloc(b->off, sizeof(double) * b->cap_reads);
        b->rng = (double *)realloc(b->rng, sizeof(double) * b->cap_reads);
        if (!b->ids || !b->offsets || !b->dig || !b->off || !b->rng) die_mem();
        if (b->n == 0) b->offsets[0] = 0;
    }
    const uint64_t o = b->offsets[b->n], n = rec->len_raw_signal;
    if (o + n > b->cap_samples) {
        uint64_t c = b->cap_samples ? b->cap_samples : (1u << 20);
        while (c < o + n) c *= 2;
        b->samples = (int16_t *)realloc(b->samples, sizeof(int16_t) * c);
        if (!b->samples) die_mem();
        b->cap_samples = c;
    }
    if (n) memcpy(b->samples + o, rec->raw_signal, sizeof(int16_t) * n); /* the record buffer is reused */
    b->ids[b->n] = strdup(rec->read_id);
    b->dig[b->n] = rec->digitisation;
    b->off[b->n] = rec->offset;
    b->rng[b->n] = rec->range;
    b->offsets[b->n + 1] = o + n;
    b->n++;
}

static void batch_clear(batch_t *b) {
    for (uint32_t i = 0; i < b->n; i++) free(b->ids[i]);
    b->n = 0;
    if (b->offsets) b->offsets[0] = 0;
}

static void batch_free(batch_t *b) {
    batch_clear(b);
    free(b->ids); free(b->offsets); free(b->dig); free(b->off); free(b->rng); free(b->samples);
    memset(b, 0, sizeof *b);
}

/* ------------------------------------------------------------------ per-GPU shard work */

typedef struct {
    int mode, device, rc;
    opt_t opt;
    const batch_t *b;
    uint32_t lo, hi; /* reads [lo, hi) of the batch */
    sgk_events_host_t ev;
    sgk_segs_host_t segs;
    sgk_stat_rec_t *stat;
    sgk_prefix_rec_t *prefix;
    float *pa;
} shard_t;

static void *shard_run(void *arg) {
    shard_t *s = (shard_t *)arg;
    const batch_t *b = s->b;
    s->rc = sgk_set_device(s->device);
    if (s->rc != SGK_OK) return NULL;
    sgk_host_batch_t hb;
    hb.samples = b->samples;
    hb.offsets = b->offsets + s->lo;
    hb.digitisation = b->dig + s->lo;
    hb.offset = b->off + s->lo;
    hb.range = b->rng + s->lo;
    hb.n_reads = s->hi - s->lo;
    const uint32_t n = hb.n_reads;
    switch (s->mode) {
        case MODE_EVENT:
            s->rc = sgk_event_host(&hb, s->opt.rna, &s->ev);
            break;
        case MODE_STAT:
            s->stat = (sgk_stat_rec_t *)calloc(n ? n : 1, sizeof(sgk_stat_rec_t));
            s->rc = s->stat ? sgk_stat_host(&hb, s->stat) : SGK_ERR_NOMEM;
            break;
        case MODE_PREFIX:
            s->prefix = (sgk_prefix_rec_t *)calloc(n ? n : 1, sizeof(sgk_prefix_rec_t));
            s->rc = s->prefix ? sgk_prefix_host(&hb, s->opt.rna, s->opt.pore, s->prefix) : SGK_ERR_NOMEM;
            break;
        case MODE_JNN:
            s->rc = sgk_jnn_host(&hb, s->opt.rna, &s->segs);
            break;
        case MODE_PA: {
            const uint64_t tot = b->offsets[s->hi]; /* sgk_pa_host indexes its output with the CSR offsets */
            s->pa = (float *)malloc(sizeof(float) * (tot ? tot : 1));
            s->rc = s->pa ? sgk_pa_host(&hb, s->pa) : SGK_ERR_NOMEM;
            break;
        }
    }
    return NULL;
}

/* ------------------------------------------------------------------ printers (src/cfunc.c) */

/* print_events, cfunc.c:16-61 */
static void print_events(const char *rid, uint64_t len, const sgk_events_host_t *ev, uint32_t r, opt_t opt) {
    const uint64_t a = ev->ev_offsets[r], n = ev->ev_offsets[r + 1] - a;
    if (opt.compact) {
        printf("%s\t%ld\t", rid, (long)len);
        if (n) {
            printf("%ld\t%ld\t", (long)ev->start[a], (long)(ev->start[a + n - 1] + ev->length[a + n - 1]));
            printf("%ld\t", (long)n);
            for (uint64_t j = 0; j < n; j++) {
                const int mi = (int)ev->length[a + j];
                if (mi) {
                    if (j < n - 1) printf("%d,", mi);
                    else printf("%d", mi);
                }
            }
        } else {
            printf(".\t.\t.\t.");
        }
        printf("\n");
    } else {
        for (uint64_t j = 0; j < n; j++)
            printf("%s\t%d\t%ld\t%ld\t%f\t%f\n", rid, (int)j, (long)ev->start[a + j],
                   (long)(ev->start[a + j] + ev->length[a + j]), ev->mean[a + j], ev->stdv[a + j]);
        printf("\n"); /* cfunc.c:58 */
    }
}

/* jnn_print, jnn.c:309-350 */
static void print_jnn(const char *rid, uint64_t len, const sgk_segs_host_t *sg, uint32_t r, opt_t opt) {
    printf("%s\t", rid);
    printf("%ld\t", (long)len);
    if (len > 0) {
        const uint64_t a = sg->seg_offsets[r], n = sg->seg_offsets[r + 1] - a;
        printf("%d\t", (int)n);
        if (opt.compact) {
            uint64_t ci = 0, mi = 0;
            for (uint64_t i = 0; i < n; i++) {
                ci += (mi = (uint64_t)sg->x[a + i] - ci);
                if (mi) printf("%dH", (int)mi);
                ci += (mi = (uint64_t)sg->y[a + i] - ci);
                if (mi) printf("%d,", (int)mi);
            }
        } else {
            for (uint64_t i = 0; i < n; i++) printf("%ld,%ld;", (long)sg->x[a + i], (long)sg->y[a + i]);
        }
        if (n == 0) printf(".");
    }
    printf("\n");
}

/* stat_func, cfunc.c:126-159 */
static void print_stat(const char *rid, uint64_t len, const sgk_stat_rec_t *s) {
    printf("%s\t", rid);
    printf("%ld\t", (long)len);
    printf("%f\t%f\t%f\t%f\t%d\t%f\t", s->raw_mean, s->pa_mean, s->raw_std, s->pa_std, (int)(int16_t)s->raw_median,
           s->pa_median);
    printf("\n");
}

/* prefix_func, cfunc.c:169-234 */
static void print_prefix(const char *rid, uint64_t len, const sgk_prefix_rec_t *p, opt_t opt) {
    printf("%s\t%ld\t", rid, (long)len);
    if (p->adapt_y > 0) {
        printf("%ld\t%ld\t", (long)p->adapt_x, (long)p->adapt_y);
        if (p->polya_y > 0) printf("%ld\t%ld", (long)p->polya_x + p->adapt_y, (long)p->polya_y + p->adapt_y);
        else printf(".\t.");
        if (opt.p_stat) {
            printf("\t%f\t%f\t%f\t", p->adapt_mean, p->adapt_std, p->adapt_median);
            if (p->polya_y > 0) printf("\t%f\t%f\t%f\t", p->polya_mean, p->polya_std, p->polya_median);
            else printf("\t.\t.\t.");
        }
    } else {
        printf(".\t.\t.\t.");
    }
    printf("\n");
}

/* pa_func, cfunc.c:85-102 */
static void print_pa(const char *rid, uint64_t len, const float *pa) {
    printf("%s\t%ld\t", rid, (long)len);
    for (uint64_t i = 0; i < len; i++) {
        if (i == len - 1) printf("%f", pa[i]);
        else printf("%f,", pa[i]);
    }
    printf("\n");
}

/* ------------------------------------------------------------------ batch processing */

static void process_batch(const batch_t *b, int mode, opt_t opt, int n_gpus) {
    if (b->n == 0) return;
    shard_t *sh = (shard_t *)calloc((size_t)n_gpus, sizeof(shard_t));
    pthread_t *th = (pthread_t *)calloc((size_t)n_gpus, sizeof(pthread_t));
    if (!sh || !th) die_mem();
    /* contiguous read ranges balanced by cumulative sample count (lengths vary widely in real data) */
    const uint64_t total = b->offsets[b->n];
    uint32_t lo = 0;
    int used = 0;
    for (int g = 0; g < n_gpus && lo < b->n; g++) {
        uint32_t hi = b->n;
        if (g < n_gpus - 1) {
            const uint64_t target = total / (uint64_t)n_gpus * (uint64_t)(g + 1);
            hi = lo;
            while (hi < b->n && b->offsets[hi + 1] <= target) hi++;
            if (hi == lo) hi = lo + 1;
        }
        sh[used].mode = mode; sh[used].device = g; sh[used].opt = opt; sh[used].b = b;
        sh[used].lo = lo; sh[used].hi = hi;
        lo = hi;
        used++;
    }
    for (int g = 1; g < used; g++) pthread_create(&th[g], NULL, shard_run, &sh[g]);
    shard_run(&sh[0]);
    for (int g = 1; g < used; g++) pthread_join(th[g], NULL);
    for (int g = 0; g < used; g++) {
        if (sh[g].rc != SGK_OK) {
            ERROR("process_batch", "GPU %d: %s %s", sh[g].device, sgk_strerror(sh[g].rc), sgk_last_hip_error());
            exit(EXIT_FAILURE);
        }
    }
    for (int g = 0; g < used; g++) {
        for (uint32_t r = sh[g].lo; r < sh[g].hi; r++) {
            const uint32_t k = r - sh[g].lo;
            const uint64_t len = b->offsets[r + 1] - b->offsets[r];
            switch (mode) {
                case MODE_EVENT: print_events(b->ids[r], len, &sh[g].ev, k, opt); break;
                case MODE_JNN: print_jnn(b->ids[r], len, &sh[g].segs, k, opt); break;
                case MODE_STAT: print_stat(b->ids[r], len, &sh[g].stat[k]); break;
                case MODE_PREFIX: print_prefix(b->ids[r], len, &sh[g].prefix[k], opt); break;
                case MODE_PA: print_pa(b->ids[r], len, sh[g].pa + b->offsets[r]); break;
            }
        }
        sgk_events_host_free(&sh[g].ev);
        sgk_segs_host_free(&sh[g].segs);
        free(sh[g].stat); free(sh[g].prefix); free(sh[g].pa);
    }
    free(th);
    free(sh);
}

/* ------------------------------------------------------------------ cmain (src/cmain.c:40-156) */

static struct option long_options[] = {
    {"verbose", required_argument, 0, 'v'}, {"help", no_argument, 0, 'h'},       {"version", no_argument, 0, 'V'},
    {"output", required_argument, 0, 'o'},  {"print-stat", no_argument, 0, 0},   {"no-header", no_argument, 0, 'n'},
    {"compact", no_argument, 0, 'c'},       {"gpus", required_argument, 0, 0},   {"batch-samples", required_argument, 0, 0},
    {0, 0, 0, 0}};

static int cmain(int argc, char *argv[], const char *mode_s) {
    const char *optstring = "o:hVnc";
    int longindex = 0, c;
    FILE *fp_help = stderr;
    int8_t hdr = 1;
    opt_t opt = {0, 0, 0, 0};
    int n_gpus = 1;
    uint64_t batch_samples = 64ull << 20;

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
        } else if (c == 0 && longindex == 4) {
            opt.p_stat = 1;
        } else if (c == 0 && longindex == 7) {
            n_gpus = atoi(optarg);
        } else if (c == 0 && longindex == 8) {
            batch_samples = strtoull(optarg, NULL, 10);
        }
    }
    if (argc - optind < 1 || fp_help == stdout) {
        fprintf(fp_help, "Usage: sigtk %s reads.blow5 read_id1 read_id2 .. \n", mode_s);
        fprintf(fp_help, "       sigtk %s reads.blow5\n", mode_s);
        fprintf(fp_help, "\nbasic options:\n");
        fprintf(fp_help, "   -h                         help\n");
        fprintf(fp_help, "   -n                         suppress header\n");
        fprintf(fp_help, "   -c                         compact output\n");
        fprintf(fp_help, "   --version                  print version\n");
        fprintf(fp_help, "   --gpus INT                 number of GPUs to shard reads across [1]\n");
        fprintf(fp_help, "   --batch-samples INT        raw samples per GPU batch [67108864]\n");
        exit(fp_help == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
    }

    b5_file_t *f = b5_open(argv[optind]);
    if (!f) {
        ERROR("cmain", "cannot open %s. ", argv[optind]);
        exit(EXIT_FAILURE);
    }
    opt.rna = drna_detect(f);
    opt.pore = pore_detect(f);

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
    } else {
        mode = MODE_PA;
        if (hdr) printf("read_id\tlen_raw_signal\tpa\n");
    }

    const int ndev = sgk_device_count();
    if (ndev <= 0) {
        ERROR("cmain", "%s", "no usable GPU: sigtk-amd has no CPU compute path");
        exit(EXIT_FAILURE);
    }
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > ndev) {
        WARNING("cmain", "--gpus %d requested but %d visible; using %d", n_gpus, ndev, ndev);
        n_gpus = ndev;
    }

    b5_rec_t rec;
    memset(&rec, 0, sizeof rec);
    batch_t b;
    memset(&b, 0, sizeof b);
    const uint64_t limit = batch_samples * (uint64_t)n_gpus;
    int ret = 0;
    if (argc - optind == 1) {
        while ((ret = b5_next(f, &rec)) >= 0) {
            batch_push(&b, &rec);
            if (b.offsets[b.n] >= limit) {
                process_batch(&b, mode, opt, n_gpus);
                batch_clear(&b);
            }
        }
        if (ret != B5_EOF) {
            fprintf(stderr, "Error in slow5_get_next. Error code %d\n", ret);
            exit(EXIT_FAILURE);
        }
    } else {
        if (b5_index(f) < 0) {
            ERROR("cmain", "Error loading index file for %s", argv[optind]);
            exit(EXIT_FAILURE);
        }
        for (int i = optind + 1; i < argc; i++) {
            fprintf(stderr, "Read ID %s\n", argv[i]);
            if (b5_get(f, argv[i], &rec) < 0) {
                ERROR("cmain", "%s", "Error when fetching the read");
                exit(EXIT_FAILURE);
            }
            batch_push(&b, &rec);
            if (b.offsets[b.n] >= limit) {
                process_batch(&b, mode, opt, n_gpus);
                batch_clear(&b);
            }
        }
    }
    process_batch(&b, mode, opt, n_gpus);
    batch_free(&b);
    b5_rec_free(&rec);
    b5_close(f);
    return 0;
}

/* hidden helper for tests: dump id, length, scaling and a checksum of every record */
static int dumpmain(int argc, char *argv[]) {
    if (argc < 2) return 1;
    b5_file_t *f = b5_open(argv[1]);
    if (!f) {
        ERROR("dumpmain", "cannot open %s. ", argv[1]);
        return 1;
    }
    b5_rec_t rec;
    memset(&rec, 0, sizeof rec);
    int ret;
    printf("#press\t%d\t%d\tgroups\t%u\n", f->record_press, f->signal_press, f->num_read_groups);
    while ((ret = b5_next(f, &rec)) >= 0) {
        uint64_t h = 1469598103934665603ull;
        for (uint64_t i = 0; i < rec.len_raw_signal; i++) {
            h ^= (uint16_t)rec.raw_signal[i];
            h *= 1099511628211ull;
        }
        printf("%s\t%lu\t%.17g\t%.17g\t%.17g\t%016lx\n", rec.read_id, (unsigned long)rec.len_raw_signal,
               rec.digitisation, rec.offset, rec.range, (unsigned long)h);
    }
    b5_rec_free(&rec);
    b5_close(f);
    return ret == B5_EOF ? 0 : 1;
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
    fprintf(fp, "\n(sigtk-amd: the per-read raw-signal subtools on MI355X; sref/ss/ent/qts are not part of it)\n");
    exit(fp == stdout ? EXIT_SUCCESS : EXIT_FAILURE);
}

int main(int argc, char *argv[]) {
    const double realtime0 = realtime();
    int ret = 1;
    if (argc < 2) {
        print_usage(stderr);
    } else if (strcmp(argv[1], "event") == 0 || strcmp(argv[1], "stat") == 0 || strcmp(argv[1], "prefix") == 0 ||
               strcmp(argv[1], "pa") == 0 || strcmp(argv[1], "jnn") == 0) {
        ret = cmain(argc - 1, argv + 1, argv[1]);
    } else if (strcmp(argv[1], "_dump") == 0) {
        return dumpmain(argc - 1, argv + 1);
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
    return ret;
}
