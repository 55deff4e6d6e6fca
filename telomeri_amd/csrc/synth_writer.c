/* telomeri_amd/csrc/synth_writer.c -- BENCH INFRASTRUCTURE (not part of the product, never loaded by it).
 *
 * Renders a synthetic dataset (telomeri_amd.synth: integer columns) as the text files the host program reads: 16-column
 * PAF (the 12 columns the reference parses + 4 it discards, OverlapGraph.cpp:109-127) and two-line FASTA records
 * (Scaffolder.cpp:12-41).  Same text as synth.write_paf / synth.write_fasta, produced by all cores: the A. thaliana-shaped
 * input is 150 M PAF lines (15 GB) and 10 GB of reads, which the Python writers would take hours to print.
 *
 *   gcc -O2 -shared -fPIC -pthread -o libsynth_writer.so synth_writer.c
 */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static char *put_uint(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char) ('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

static char *put_int(char *p, int64_t v) {
    if (v < 0) { *p++ = '-'; return put_uint(p, (uint64_t) (-v)); }
    return put_uint(p, (uint64_t) v);
}

/* synth.Dataset.name_of: reads are "read%07d" (1-based), contigs "ctg%d" (1-based) */
static char *put_name(char *p, int64_t id, int64_t n_reads) {
    if (id < n_reads) {
        memcpy(p, "read", 4); p += 4;
        uint64_t v = (uint64_t) id + 1;
        char tmp[24];
        int n = 0;
        do { tmp[n++] = (char) ('0' + v % 10); v /= 10; } while (v);
        for (int k = n; k < 7; k++) *p++ = '0';
        while (n) *p++ = tmp[--n];
        return p;
    }
    memcpy(p, "ctg", 3); p += 3;
    return put_uint(p, (uint64_t) (id - n_reads + 1));
}

typedef struct {
    const int32_t *q_id, *t_id, *q_len, *q_start, *q_end, *t_len, *t_start, *t_end, *matches, *strand;
    int64_t n_reads;
    uint64_t lo, hi;
    char *buf;
    size_t len;
} paf_job;

#define PAF_LINE_MAX 256

static void *paf_worker(void *arg) {
    paf_job *j = (paf_job *) arg;
    j->buf = (char *) malloc((j->hi - j->lo) * PAF_LINE_MAX + 16);
    if (!j->buf) return NULL;
    char *p = j->buf;
    for (uint64_t i = j->lo; i < j->hi; i++) {
        const int64_t ovl = (int64_t) j->q_end[i] - (int64_t) j->q_start[i];
        p = put_name(p, j->q_id[i], j->n_reads); *p++ = '\t';
        p = put_int(p, j->q_len[i]); *p++ = '\t';
        p = put_int(p, j->q_start[i]); *p++ = '\t';
        p = put_int(p, j->q_end[i]); *p++ = '\t';
        *p++ = j->strand[i] ? '-' : '+'; *p++ = '\t';
        p = put_name(p, j->t_id[i], j->n_reads); *p++ = '\t';
        p = put_int(p, j->t_len[i]); *p++ = '\t';
        p = put_int(p, j->t_start[i]); *p++ = '\t';
        p = put_int(p, j->t_end[i]); *p++ = '\t';
        p = put_int(p, j->matches[i]); *p++ = '\t';
        p = put_int(p, ovl); *p++ = '\t';
        memcpy(p, "60\ttp:A:S\tcm:i:", 15); p += 15;
        /* Python's floor division */
        p = put_int(p, ovl >= 0 ? ovl / 4 : -((-ovl + 3) / 4));
        memcpy(p, "\ts1:i:", 6); p += 6;
        p = put_int(p, j->matches[i]);
        memcpy(p, "\tdv:f:0.0100\n", 13); p += 13;
    }
    j->len = (size_t) (p - j->buf);
    return NULL;
}

static int write_all(int fd, const char *p, size_t n) {
    while (n) {
        ssize_t w = write(fd, p, n > (1u << 30) ? (1u << 30) : n);
        if (w <= 0) return -1;
        p += w;
        n -= (size_t) w;
    }
    return 0;
}

/* Returns bytes written, or -1. */
int64_t synth_write_paf(const char *path, uint64_t n, const int32_t *q_id, const int32_t *t_id, const int32_t *q_len,
                        const int32_t *q_start, const int32_t *q_end, const int32_t *t_len, const int32_t *t_start,
                        const int32_t *t_end, const int32_t *matches, const int32_t *strand, int64_t n_reads, int threads) {
    int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return -1;
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    int64_t total = 0;
    /* slabs of 4 M records per round keep the staging memory bounded (4 M x 256 B = 1 GB worst case per round) */
    const uint64_t slab = 4u << 20;
    for (uint64_t base = 0; base < n; base += slab) {
        const uint64_t end = base + slab < n ? base + slab : n;
        paf_job jobs[64];
        pthread_t th[64];
        for (int t = 0; t < threads; t++) {
            paf_job *j = &jobs[t];
            j->q_id = q_id; j->t_id = t_id; j->q_len = q_len; j->q_start = q_start; j->q_end = q_end;
            j->t_len = t_len; j->t_start = t_start; j->t_end = t_end; j->matches = matches; j->strand = strand;
            j->n_reads = n_reads;
            j->lo = base + (end - base) * (uint64_t) t / (uint64_t) threads;
            j->hi = base + (end - base) * (uint64_t) (t + 1) / (uint64_t) threads;
            j->buf = NULL;
            j->len = 0;
            pthread_create(&th[t], NULL, paf_worker, j);
        }
        int bad = 0;
        for (int t = 0; t < threads; t++) {
            pthread_join(th[t], NULL);
            if (!jobs[t].buf || write_all(fd, jobs[t].buf, jobs[t].len)) bad = 1;
            total += (int64_t) jobs[t].len;
            free(jobs[t].buf);
        }
        if (bad) { close(fd); return -1; }
    }
    close(fd);
    return total;
}

/* Two-line records ">name\nSEQ\n" cut out of `genome` (forward strand): record i = genome[start[i], start[i] + len[i]).
 * Names are name_of(first_id + i). */
int64_t synth_write_fasta(const char *path, const uint8_t *genome, uint64_t n, const int64_t *start, const int64_t *len,
                          int64_t first_id, int64_t n_reads) {
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    static char big[1 << 22];
    setvbuf(f, big, _IOFBF, sizeof(big));
    int64_t total = 0;
    char name[64];
    for (uint64_t i = 0; i < n; i++) {
        char *p = name;
        *p++ = '>';
        p = put_name(p, first_id + (int64_t) i, n_reads);
        *p++ = '\n';
        fwrite(name, 1, (size_t) (p - name), f);
        fwrite(genome + start[i], 1, (size_t) len[i], f);
        fputc('\n', f);
        total += (p - name) + len[i] + 1;
    }
    if (fclose(f)) return -1;
    return total;
}
