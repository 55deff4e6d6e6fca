/* oracle/telomeri_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See telomeri_oracle.h.
 *
 * Every function cites the reference lines it restates.  Nothing here is reached from the
 * product path (telomeri_amd/ and include/): that path fails loudly without the HIP library.
 */
#define _GNU_SOURCE
#include "telomeri_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* std::mt19937 (libstdc++ mersenne_twister_engine<uint32,32,624,397,31,0x9908b0df,11,0xffffffff,
 * 7,0x9d2c5680,15,0xefc60000,18,1812433253>) and generate_canonical<double,53>               */

typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937;

static void mt_seed(mt19937 *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
    g->idx = 624;
}

static void mt_twist(mt19937 *g) {
    uint32_t *mt = g->mt;
    for (int i = 0; i < 624; i++) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    g->idx = 0;
}

static uint32_t mt_next(mt19937 *g) {
    if (g->idx >= 624) mt_twist(g);
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* uniform_real_distribution<double>(0,1)(gen) == generate_canonical<double,53>(gen): two 32-bit
 * draws, low word first; sum = x0 + x1*2^32 rounded once to double; / 2^64; clamp below 1
 * (bits/random.tcc:3348-3380, bits/random.h:1870). PathManager.cpp:13-14,104. */
static double mt_canonical(mt19937 *g) {
    double x0 = (double) mt_next(g);
    double x1 = (double) mt_next(g);
    double sum = x0 + x1 * 4294967296.0;
    double ret = sum / 18446744073709551616.0;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}

void tlo_mt19937_raw(uint32_t seed, size_t n, uint32_t *out) {
    mt19937 g;
    mt_seed(&g, seed);
    for (size_t i = 0; i < n; i++) out[i] = mt_next(&g);
}

void tlo_mt19937_canonical(uint32_t seed, size_t n, double *out) {
    mt19937 g;
    mt_seed(&g, seed);
    for (size_t i = 0; i < n; i++) out[i] = mt_canonical(&g);
}

/* Per-walk seed (seed_mode PER_WALK; the device uses the same function, include/telomeri_hip.h). */
uint32_t tlo_walk_seed(int metric, uint32_t anchor_node, uint32_t r) {
    return 5489u + 0x9E3779B9u * (anchor_node + 1u) + 0x85EBCA6Bu * r + 0xC2B2AE35u * (uint32_t) metric;
}

/* ------------------------------------------------------------------------------------------ */
/* a2: OverlapGraph::filter (OverlapGraph.cpp:156-202) and a3: scores (OverlapGraph.cpp:221-231,297-324) */

static int filter_one(const tlo_filter_params *p, int q_len, int q_start, int q_end, int t_len, int t_start,
                      int t_end, int same_name) {
    if (same_name) return 0;                                 /* :157 */
    int qol = q_end - q_start;                               /* :297-299 */
    int tol = t_end - t_start;                               /* :301-303 */
    int qoh = q_len - q_end;                                 /* :305-307 */
    int toh = t_start;                                       /* :309-311 */
    float overlap_length, overhang_length, total_length;
    switch (p->mode) {                                       /* :170-193 */
        case TLO_MODE_MIN:
            overlap_length = (float) (qol < tol ? qol : tol);
            overhang_length = (float) (qoh < toh ? qoh : toh);
            total_length = (float) (q_len < t_len ? q_len : t_len);
            break;
        case TLO_MODE_MAX:
            overlap_length = (float) (qol > tol ? qol : tol);
            overhang_length = (float) (qoh > toh ? qoh : toh);
            total_length = (float) (q_len > t_len ? q_len : t_len);
            break;
        case TLO_MODE_AVG:
            overlap_length = (float) (qol + tol) / 2.0f;
            overhang_length = (float) (qoh + toh) / 2.0f;
            total_length = (float) (q_len + t_len) / 2.0f;
            break;
        case TLO_MODE_SUM:
            overlap_length = (float) (qol + tol);
            overhang_length = (float) (qoh + toh);
            total_length = (float) (q_len + t_len);
            break;
        default:
            return 0;
    }
    /* :195-198; the ulong thresholds are converted to float by the usual arithmetic conversions */
    if (overlap_length < (float) p->min_overlap_length
        || (overlap_length / total_length) < p->min_overlap_percentage)
        return 0;
    /* :200-201 */
    return overhang_length <= (float) p->max_overhang_length
           && (overhang_length / overlap_length) <= p->max_overhang_percentage;
}

static void score_one(int q_len, int q_start, int q_end, int t_len, int t_start, int t_end, int matches, float *si,
                      float *os, float *es) {
    int qol = q_end - q_start, tol = t_end - t_start;
    int qel = q_start;                                       /* :313-315 */
    int qoh = q_len - q_end, toh = t_start;
    float SI = (float) matches / (float) (q_len < t_len ? q_len : t_len);   /* :321-324 */
    float OS = (float) (qol + tol) / 2.0f * SI;                                /* :228 */
    float ES = fabsf(OS + (float) qel / 2.0f - (float) (qoh + toh) / 2.0f);    /* :231 */
    *si = SI;
    *os = OS;
    *es = ES;
}

void tlo_filter_score(const tlo_filter_params *p, size_t n, const int32_t *q_len, const int32_t *q_start,
                      const int32_t *q_end, const int32_t *t_len, const int32_t *t_start, const int32_t *t_end,
                      const int32_t *matches, const uint8_t *same_name, uint8_t *keep, float *si, float *os,
                      float *es) {
    for (size_t i = 0; i < n; i++) {
        keep[i] = (uint8_t) filter_one(p, q_len[i], q_start[i], q_end[i], t_len[i], t_start[i], t_end[i],
                                       same_name ? same_name[i] : 0);
        score_one(q_len[i], q_start[i], q_end[i], t_len[i], t_start[i], t_end[i], matches[i], &si[i], &os[i], &es[i]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a4: graph (OverlapGraph.cpp:204-258) */

struct tlo_graph {
    uint32_t n_nodes, cap_nodes;
    uint32_t *node_len;
    uint8_t *node_anchor;
    char **node_name;            /* text mode */
    int64_t *node_ext;           /* column mode */
    uint32_t *htab;              /* open addressing, node index + 1 */
    uint32_t hcap;
    int32_t *ext2node;
    size_t ext_cap;
    /* directed adjacency entries in arrival order (two per kept record) */
    uint64_t n_ent, cap_ent;
    uint32_t *e_owner, *e_nbr, *e_qs, *e_qe, *e_ts, *e_te;
    float *e_os, *e_si, *e_es;
    uint8_t *e_strand;
    uint64_t n_edges;
    /* CSR */
    int csr_valid;
    uint64_t *row_ptr;
    uint32_t *c_nbr, *c_qs, *c_qe, *c_ts, *c_te;
    float *c_os, *c_si, *c_es;
    uint8_t *c_strand;
    /* anchors in node order; DET unit table */
    uint32_t n_anchors;
    uint32_t *anchors;
    uint64_t *det_off;           /* [n_anchors+1] prefix of DET units per anchor */
};

tlo_graph *tlo_graph_new(void) { return (tlo_graph *) calloc(1, sizeof(tlo_graph)); }

static void free_csr(tlo_graph *g) {
    free(g->row_ptr); free(g->c_nbr); free(g->c_qs); free(g->c_qe); free(g->c_ts); free(g->c_te);
    free(g->c_os); free(g->c_si); free(g->c_es); free(g->c_strand); free(g->anchors); free(g->det_off);
    g->row_ptr = NULL; g->c_nbr = g->c_qs = g->c_qe = g->c_ts = g->c_te = NULL;
    g->c_os = g->c_si = g->c_es = NULL; g->c_strand = NULL; g->anchors = NULL; g->det_off = NULL;
    g->csr_valid = 0;
}

void tlo_graph_free(tlo_graph *g) {
    if (!g) return;
    free_csr(g);
    if (g->node_name) for (uint32_t i = 0; i < g->n_nodes; i++) free(g->node_name[i]);
    free(g->node_name); free(g->node_ext); free(g->node_len); free(g->node_anchor); free(g->htab); free(g->ext2node);
    free(g->e_owner); free(g->e_nbr); free(g->e_qs); free(g->e_qe); free(g->e_ts); free(g->e_te);
    free(g->e_os); free(g->e_si); free(g->e_es); free(g->e_strand);
    free(g);
}

static uint32_t fnv1a(const char *s, size_t n) {
    uint32_t h = 2166136261u;
    for (size_t i = 0; i < n; i++) { h ^= (uint8_t) s[i]; h *= 16777619u; }
    return h;
}

static void grow_nodes(tlo_graph *g) {
    if (g->n_nodes < g->cap_nodes) return;
    uint32_t nc = g->cap_nodes ? g->cap_nodes * 2 : 1024;
    g->node_len = (uint32_t *) realloc(g->node_len, nc * sizeof(uint32_t));
    g->node_anchor = (uint8_t *) realloc(g->node_anchor, nc);
    g->node_name = (char **) realloc(g->node_name, nc * sizeof(char *));
    g->node_ext = (int64_t *) realloc(g->node_ext, nc * sizeof(int64_t));
    g->cap_nodes = nc;
}

static uint32_t new_node(tlo_graph *g, int anchor, uint32_t length, const char *name, size_t name_len, int64_t ext) {
    grow_nodes(g);
    uint32_t i = g->n_nodes++;
    g->node_len[i] = length;
    g->node_anchor[i] = (uint8_t) (anchor != 0);
    g->node_name[i] = NULL;
    g->node_ext[i] = ext;
    if (name) {
        g->node_name[i] = (char *) malloc(name_len + 1);
        memcpy(g->node_name[i], name, name_len);
        g->node_name[i][name_len] = 0;
    }
    return i;
}

static void hrehash(tlo_graph *g) {
    uint32_t nc = g->hcap ? g->hcap * 2 : 4096;
    uint32_t *nt = (uint32_t *) calloc(nc, sizeof(uint32_t));
    for (uint32_t i = 0; i < g->n_nodes; i++) {
        if (!g->node_name[i]) continue;
        uint32_t h = fnv1a(g->node_name[i], strlen(g->node_name[i])) & (nc - 1);
        while (nt[h]) h = (h + 1) & (nc - 1);
        nt[h] = i + 1;
    }
    free(g->htab);
    g->htab = nt;
    g->hcap = nc;
}

/* nodeIndex (OverlapGraph.cpp:251-258): names are unique in nodes_, so "last match" == "the match". */
static long node_index_by_name(tlo_graph *g, const char *s, size_t n) {
    if (!g->hcap) return -1;
    uint32_t h = fnv1a(s, n) & (g->hcap - 1);
    while (g->htab[h]) {
        uint32_t i = g->htab[h] - 1;
        if (strlen(g->node_name[i]) == n && memcmp(g->node_name[i], s, n) == 0) return (long) i;
        h = (h + 1) & (g->hcap - 1);
    }
    return -1;
}

static void hinsert(tlo_graph *g, uint32_t node) {
    if ((uint64_t) g->n_nodes * 2 > g->hcap) hrehash(g);  /* rehash already includes `node` */
    else {
        uint32_t h = fnv1a(g->node_name[node], strlen(g->node_name[node])) & (g->hcap - 1);
        while (g->htab[h]) h = (h + 1) & (g->hcap - 1);
        g->htab[h] = node + 1;
    }
}

static void grow_ent(tlo_graph *g, uint64_t need) {
    if (g->n_ent + need <= g->cap_ent) return;
    uint64_t nc = g->cap_ent ? g->cap_ent * 2 : 8192;
    while (nc < g->n_ent + need) nc *= 2;
#define RE(f, T) g->f = (T *) realloc(g->f, nc * sizeof(T))
    RE(e_owner, uint32_t); RE(e_nbr, uint32_t); RE(e_qs, uint32_t); RE(e_qe, uint32_t); RE(e_ts, uint32_t);
    RE(e_te, uint32_t); RE(e_os, float); RE(e_si, float); RE(e_es, float); RE(e_strand, uint8_t);
#undef RE
    g->cap_ent = nc;
}

/* The tail of buildFrom (OverlapGraph.cpp:233-248): edges_.push_back, then the entry owned by the
 * target (unswapped) and the entry owned by the query (q/t index, start, end swapped; scores and
 * strand NOT recomputed). */
static void add_edge(tlo_graph *g, uint32_t qn, uint32_t tn, int q_start, int q_end, int t_start, int t_end, float os,
                     float si, float es, int minus) {
    grow_ent(g, 2);
    uint64_t k = g->n_ent;
    g->e_owner[k] = tn; g->e_nbr[k] = qn;
    g->e_qs[k] = (uint32_t) q_start; g->e_qe[k] = (uint32_t) q_end;
    g->e_ts[k] = (uint32_t) t_start; g->e_te[k] = (uint32_t) t_end;
    g->e_os[k] = os; g->e_si[k] = si; g->e_es[k] = es; g->e_strand[k] = (uint8_t) minus;
    k++;
    g->e_owner[k] = qn; g->e_nbr[k] = tn;
    g->e_qs[k] = (uint32_t) t_start; g->e_qe[k] = (uint32_t) t_end;
    g->e_ts[k] = (uint32_t) q_start; g->e_te[k] = (uint32_t) q_end;
    g->e_os[k] = os; g->e_si[k] = si; g->e_es[k] = es; g->e_strand[k] = (uint8_t) minus;
    g->n_ent += 2;
    g->n_edges++;
    g->csr_valid = 0;
}

/* Utils::startsWithInsensitive(name, "ctg") (Utils.hpp:17-27) */
static int starts_with_ctg(const char *s, size_t n) {
    if (n < 3) return 0;
    return (s[0] == 'c' || s[0] == 'C') && (s[1] == 't' || s[1] == 'T') && (s[2] == 'g' || s[2] == 'G');
}

/* --- text parsing with operator>> semantics (OverlapGraph.cpp:109-127) ----------------------- */
typedef struct { const char *p, *end; } rd;

static int is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
static void skip_ws(rd *r) { while (r->p < r->end && is_ws(*r->p)) r->p++; }

static int rd_token(rd *r, const char **s, size_t *n) {   /* >> std::string */
    skip_ws(r);
    if (r->p >= r->end) return 0;
    *s = r->p;
    while (r->p < r->end && !is_ws(*r->p)) r->p++;
    *n = (size_t) (r->p - *s);
    return 1;
}

static int rd_int(rd *r, int *v) {                         /* >> int : [+-]digits, failbit on overflow */
    skip_ws(r);
    if (r->p >= r->end) return 0;
    int neg = 0;
    const char *q = r->p;
    if (*q == '+' || *q == '-') { neg = (*q == '-'); q++; }
    const char *d0 = q;
    long long acc = 0;
    int ovf = 0;
    while (q < r->end && *q >= '0' && *q <= '9') {
        if (!ovf) { acc = acc * 10 + (*q - '0'); if (acc > 4294967296LL) ovf = 1; }
        q++;
    }
    if (q == d0) return 0;
    r->p = q;
    if (neg) acc = -acc;
    if (ovf || acc > 2147483647LL || acc < -2147483648LL) return 0;
    *v = (int) acc;
    return 1;
}

static int rd_char(rd *r, char *c) {                       /* >> char */
    skip_ws(r);
    if (r->p >= r->end) return 0;
    *c = *r->p++;
    return 1;
}

int tlo_graph_load_paf(tlo_graph *g, const char *path, int anchors, const tlo_filter_params *p,
                       uint32_t test_load_num, uint64_t *records_out) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "Cannot open file: %s\n", path); return -1; }   /* :100-103 */
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *) malloc(sz > 0 ? (size_t) sz : 1);
    size_t got = sz > 0 ? fread(buf, 1, (size_t) sz, f) : 0;
    fclose(f);
    rd r = {buf, buf + got};
    uint64_t ctr = 0;
    for (;;) {
        const char *qn, *tn, *dummy;
        size_t qn_n, tn_n, dn;
        int q_len, q_start, q_end, t_len, t_start, t_end, matches, blen, mapq;
        char strand;
        if (!(rd_token(&r, &qn, &qn_n) && rd_int(&r, &q_len) && rd_int(&r, &q_start) && rd_int(&r, &q_end)
              && rd_char(&r, &strand) && rd_token(&r, &tn, &tn_n) && rd_int(&r, &t_len) && rd_int(&r, &t_start)
              && rd_int(&r, &t_end) && rd_int(&r, &matches) && rd_int(&r, &blen) && rd_int(&r, &mapq)))
            break;
        for (int k = 0; k < 4; k++) rd_token(&r, &dummy, &dn);   /* :124-127 */
        ++ctr;
        int pos_query = anchors && starts_with_ctg(qn, qn_n);      /* :130-138 */
        int pos_target = anchors && !pos_query;
        int same = (qn_n == tn_n && memcmp(qn, tn, qn_n) == 0);
        if (filter_one(p, q_len, q_start, q_end, t_len, t_start, t_end, same)) {
            long qi = node_index_by_name(g, qn, qn_n);             /* :208-212 */
            if (qi < 0) { qi = new_node(g, pos_query, (uint32_t) q_len, qn, qn_n, -1); hinsert(g, (uint32_t) qi); }
            long ti = node_index_by_name(g, tn, tn_n);             /* :214-218 */
            if (ti < 0) { ti = new_node(g, pos_target, (uint32_t) t_len, tn, tn_n, -1); hinsert(g, (uint32_t) ti); }
            float si, os, es;
            score_one(q_len, q_start, q_end, t_len, t_start, t_end, matches, &si, &os, &es);
            add_edge(g, (uint32_t) qi, (uint32_t) ti, q_start, q_end, t_start, t_end, os, si, es, strand == '-');
        }
        if (test_load_num > 0 && test_load_num < ctr) break;       /* :145-147 */
    }
    free(buf);
    if (records_out) *records_out = ctr;
    return 0;
}

int tlo_graph_add_records(tlo_graph *g, int anchors, const tlo_filter_params *p, size_t n, const int32_t *q_id,
                          const int32_t *t_id, const uint8_t *name_is_ctg, const int32_t *q_len,
                          const int32_t *q_start, const int32_t *q_end, const int32_t *t_len,
                          const int32_t *t_start, const int32_t *t_end, const int32_t *matches,
                          const int32_t *strand) {
    for (size_t i = 0; i < n; i++) {
        int32_t mx = q_id[i] > t_id[i] ? q_id[i] : t_id[i];
        if (q_id[i] < 0 || t_id[i] < 0) return -1;
        if ((size_t) mx >= g->ext_cap) {
            size_t nc = g->ext_cap ? g->ext_cap : 1024;
            while (nc <= (size_t) mx) nc *= 2;
            g->ext2node = (int32_t *) realloc(g->ext2node, nc * sizeof(int32_t));
            for (size_t k = g->ext_cap; k < nc; k++) g->ext2node[k] = -1;
            g->ext_cap = nc;
        }
        int pos_query = anchors && name_is_ctg && name_is_ctg[q_id[i]];
        int pos_target = anchors && !pos_query;
        if (!filter_one(p, q_len[i], q_start[i], q_end[i], t_len[i], t_start[i], t_end[i], q_id[i] == t_id[i]))
            continue;
        if (g->ext2node[q_id[i]] < 0) g->ext2node[q_id[i]] = (int32_t) new_node(g, pos_query, (uint32_t) q_len[i], NULL, 0, q_id[i]);
        if (g->ext2node[t_id[i]] < 0) g->ext2node[t_id[i]] = (int32_t) new_node(g, pos_target, (uint32_t) t_len[i], NULL, 0, t_id[i]);
        float si, os, es;
        score_one(q_len[i], q_start[i], q_end[i], t_len[i], t_start[i], t_end[i], matches[i], &si, &os, &es);
        add_edge(g, (uint32_t) g->ext2node[q_id[i]], (uint32_t) g->ext2node[t_id[i]], q_start[i], q_end[i], t_start[i],
                 t_end[i], os, si, es, strand[i] != 0);
    }
    return 0;
}

static void build_anchor_tables(tlo_graph *g) {
    free(g->anchors); free(g->det_off);
    g->n_anchors = 0;
    for (uint32_t i = 0; i < g->n_nodes; i++) g->n_anchors += g->node_anchor[i];
    g->anchors = (uint32_t *) malloc((g->n_anchors + 1) * sizeof(uint32_t));
    g->det_off = (uint64_t *) malloc((g->n_anchors + 1) * sizeof(uint64_t));
    uint32_t k = 0;
    uint64_t off = 0;
    for (uint32_t i = 0; i < g->n_nodes; i++) {
        if (!g->node_anchor[i]) continue;
        g->anchors[k] = i;
        g->det_off[k] = off;
        /* first-edge loop (PathManager.cpp:204) ends after the first edge that leads to an anchor (:224-229) */
        uint64_t cnt = 0;
        for (uint64_t e = g->row_ptr[i]; e < g->row_ptr[i + 1]; e++) {
            cnt++;
            if (g->node_anchor[g->c_nbr[e]]) break;
        }
        off += cnt;
        k++;
    }
    g->det_off[k] = off;
}

/* Stable counting sort of the arrival-order entries by owner: per-node order == record order,
 * exactly what nodes_[i].edges.push_back produces (OverlapGraph.cpp:243,248). */
static int finalize(tlo_graph *g) {
    if (g->csr_valid) return 0;
    free_csr(g);
    uint32_t N = g->n_nodes;
    uint64_t A = g->n_ent;
    g->row_ptr = (uint64_t *) calloc((size_t) N + 2, sizeof(uint64_t));
    for (uint64_t k = 0; k < A; k++) g->row_ptr[g->e_owner[k] + 1]++;
    for (uint32_t i = 0; i < N; i++) g->row_ptr[i + 1] += g->row_ptr[i];
    uint64_t *cur = (uint64_t *) malloc(((size_t) N + 1) * sizeof(uint64_t));
    memcpy(cur, g->row_ptr, ((size_t) N + 1) * sizeof(uint64_t));
    size_t a = A ? A : 1;
    g->c_nbr = (uint32_t *) malloc(a * 4); g->c_qs = (uint32_t *) malloc(a * 4); g->c_qe = (uint32_t *) malloc(a * 4);
    g->c_ts = (uint32_t *) malloc(a * 4); g->c_te = (uint32_t *) malloc(a * 4);
    g->c_os = (float *) malloc(a * 4); g->c_si = (float *) malloc(a * 4); g->c_es = (float *) malloc(a * 4);
    g->c_strand = (uint8_t *) malloc(a);
    for (uint64_t k = 0; k < A; k++) {
        uint64_t d = cur[g->e_owner[k]]++;
        g->c_nbr[d] = g->e_nbr[k]; g->c_qs[d] = g->e_qs[k]; g->c_qe[d] = g->e_qe[k]; g->c_ts[d] = g->e_ts[k];
        g->c_te[d] = g->e_te[k]; g->c_os[d] = g->e_os[k]; g->c_si[d] = g->e_si[k]; g->c_es[d] = g->e_es[k];
        g->c_strand[d] = g->e_strand[k];
    }
    free(cur);
    build_anchor_tables(g);
    g->csr_valid = 1;
    return 0;
}

int tlo_graph_from_csr(tlo_graph *g, uint32_t n_nodes, uint64_t n_adj, const uint64_t *row_ptr, const uint32_t *nbr,
                       const uint32_t *q_start, const uint32_t *t_start, const float *os, const float *es,
                       const uint32_t *node_len, const uint8_t *node_anchor) {
    if (g->n_nodes || g->n_ent) return -1;
    free_csr(g);
    g->n_nodes = g->cap_nodes = n_nodes;
    g->node_len = (uint32_t *) malloc(((size_t) n_nodes + 1) * 4);
    g->node_anchor = (uint8_t *) malloc((size_t) n_nodes + 1);
    g->node_name = (char **) calloc((size_t) n_nodes + 1, sizeof(char *));
    g->node_ext = (int64_t *) malloc(((size_t) n_nodes + 1) * 8);
    memcpy(g->node_len, node_len, (size_t) n_nodes * 4);
    memcpy(g->node_anchor, node_anchor, n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) g->node_ext[i] = -1;
    size_t a = n_adj ? n_adj : 1;
    g->row_ptr = (uint64_t *) malloc(((size_t) n_nodes + 1) * 8);
    memcpy(g->row_ptr, row_ptr, ((size_t) n_nodes + 1) * 8);
    g->c_nbr = (uint32_t *) malloc(a * 4); g->c_qs = (uint32_t *) malloc(a * 4); g->c_ts = (uint32_t *) malloc(a * 4);
    g->c_os = (float *) malloc(a * 4); g->c_es = (float *) malloc(a * 4);
    g->c_qe = (uint32_t *) calloc(a, 4); g->c_te = (uint32_t *) calloc(a, 4); g->c_si = (float *) calloc(a, 4);
    g->c_strand = (uint8_t *) calloc(a, 1);
    memcpy(g->c_nbr, nbr, n_adj * 4); memcpy(g->c_qs, q_start, n_adj * 4); memcpy(g->c_ts, t_start, n_adj * 4);
    memcpy(g->c_os, os, n_adj * 4); memcpy(g->c_es, es, n_adj * 4);
    g->n_ent = n_adj;
    g->n_edges = n_adj / 2;
    build_anchor_tables(g);
    g->csr_valid = 1;
    return 0;
}

int tlo_graph_csr(tlo_graph *g, tlo_csr *out) {
    if (finalize(g)) return -1;
    out->n_nodes = g->n_nodes; out->n_adj = g->n_ent; out->row_ptr = g->row_ptr; out->nbr = g->c_nbr;
    out->q_start = g->c_qs; out->q_end = g->c_qe; out->t_start = g->c_ts; out->t_end = g->c_te;
    out->os = g->c_os; out->si = g->c_si; out->es = g->c_es; out->strand = g->c_strand;
    out->node_len = g->node_len; out->node_anchor = g->node_anchor; out->n_edges = g->n_edges;
    return 0;
}

const char *tlo_graph_node_name(tlo_graph *g, uint32_t node) {
    if (node >= g->n_nodes || !g->node_name[node]) return "";
    return g->node_name[node];
}

int64_t tlo_graph_node_ext_id(tlo_graph *g, uint32_t node) {
    if (node >= g->n_nodes) return -1;
    return g->node_ext[node];
}

uint64_t tlo_mc_units(tlo_graph *g, const tlo_walk_params *p) {
    if (finalize(g)) return 0;
    return (uint64_t) g->n_anchors * (uint64_t) (p->rebuild_attempts > 0 ? p->rebuild_attempts : 0);
}

uint64_t tlo_det_units(tlo_graph *g) {
    if (finalize(g)) return 0;
    return g->det_off[g->n_anchors];
}

/* ------------------------------------------------------------------------------------------ */
/* walks */

/* a5: Utils::getMetric (Utils.hpp:47-60) */
static inline float metric_of(const tlo_graph *g, int metric, uint64_t e) {
    switch (metric) {
        case TLO_METRIC_ES: return g->c_es[e];
        case TLO_METRIC_ES_SQRT: return sqrtf(g->c_es[e]);
        case TLO_METRIC_OS: return g->c_os[e];
        case TLO_METRIC_OS_SQRT: return sqrtf(g->c_os[e]);
        default: return 0.0f;
    }
}

typedef struct {
    uint8_t status;
    int64_t length;
    uint32_t n_nodes;
    uint32_t *nodes;     /* malloc'ed for accepted paths only */
    uint32_t *slots;
} unit_result;

typedef struct {
    const tlo_graph *g;
    uint8_t *visited;
    uint32_t *touched;
    uint64_t n_touched;
    uint32_t *p_nodes;       /* path: nodes */
    uint64_t *p_edges;       /* path: global CSR index of each edge */
    uint64_t *cand;          /* MC candidate list / DET sort scratch */
    uint64_t *tmp;
    uint64_t cap_cand;
} scratch;

static void scratch_init(scratch *s, const tlo_graph *g) {
    uint64_t maxdeg = 1;
    for (uint32_t i = 0; i < g->n_nodes; i++) {
        uint64_t d = g->row_ptr[i + 1] - g->row_ptr[i];
        if (d > maxdeg) maxdeg = d;
    }
    s->g = g;
    s->visited = (uint8_t *) calloc((size_t) g->n_nodes + 1, 1);
    s->touched = (uint32_t *) malloc(((size_t) g->n_nodes + 2) * 4 * 2);
    s->n_touched = 0;
    s->p_nodes = (uint32_t *) malloc(((size_t) g->n_nodes + 2) * 4);
    s->p_edges = (uint64_t *) malloc(((size_t) g->n_nodes + 2) * 8);
    s->cand = (uint64_t *) malloc(maxdeg * 8);
    s->tmp = (uint64_t *) malloc(maxdeg * 8);
    s->cap_cand = maxdeg;
}

static void scratch_free(scratch *s) {
    free(s->visited); free(s->touched); free(s->p_nodes); free(s->p_edges); free(s->cand); free(s->tmp);
}

static inline void mark(scratch *s, uint32_t n) {
    if (!s->visited[n]) { s->visited[n] = 1; s->touched[s->n_touched++] = n; }
}

static inline void clear_visited(scratch *s) {
    for (uint64_t i = 0; i < s->n_touched; i++) s->visited[s->touched[i]] = 0;
    s->n_touched = 0;
}

static inline int64_t edge_delta(const tlo_graph *g, uint64_t e) {
    return (int64_t) g->c_qs[e] - (int64_t) g->c_ts[e];     /* Path.cpp:14 */
}

/* Path::updateLength (Path.cpp:6-16): nodes_[0]->length + sum over all edges but the last. */
static inline int64_t path_length(const tlo_graph *g, const scratch *s, uint32_t m_edges, int64_t sum_all) {
    int64_t len = (int64_t) g->node_len[s->p_nodes[0]];
    if (m_edges >= 1) len += sum_all - edge_delta(g, s->p_edges[m_edges - 1]);
    return len;
}

static void store_path(const tlo_graph *g, const scratch *s, uint32_t m_edges, unit_result *res) {
    res->n_nodes = m_edges + 1;
    res->nodes = (uint32_t *) malloc(((size_t) m_edges + 1) * 4);
    res->slots = (uint32_t *) malloc(((size_t) m_edges + 1) * 4);
    for (uint32_t i = 0; i <= m_edges; i++) res->nodes[i] = s->p_nodes[i];
    for (uint32_t i = 0; i < m_edges; i++) res->slots[i] = (uint32_t) (s->p_edges[i] - g->row_ptr[s->p_nodes[i]]);
    res->slots[m_edges] = 0xFFFFFFFFu;
}

/* One Monte-Carlo trial (PathManager.cpp:30-160). */
static void mc_walk(scratch *s, int metric, const tlo_walk_params *p, uint32_t start, mt19937 *gen, unit_result *res,
                    tlo_unit_stat *st) {
    const tlo_graph *g = s->g;
    uint32_t n = start;
    uint32_t m = 0;                      /* p.edges_.size() */
    int64_t sum_all = 0;
    int acceptable = 0;
    int backtracks = 0;
    int64_t len = 0;
    s->p_nodes[0] = start;               /* :39 */
    mark(s, start);                      /* :33 */
    memset(st, 0, sizeof(*st));
    for (;;) {                           /* :48 */
        double sum = 0;
        int64_t anchor_edge = -1;
        uint64_t ncand = 0;
        for (uint64_t e = g->row_ptr[n]; e < g->row_ptr[n + 1]; e++) {     /* :53 */
            uint32_t q = g->c_nbr[e];
            st->evals++;
            if (g->node_anchor[q] && q != start) { anchor_edge = (int64_t) e; break; }   /* :57-60 */
            if (!s->visited[q]) { sum += (double) metric_of(g, metric, e); s->cand[ncand++] = e; }  /* :63-66 */
        }
        st->steps++;
        int64_t edge = -1;
        if (anchor_edge >= 0) {
            edge = anchor_edge;          /* :70-71 */
        } else if (ncand == 0) {         /* :72 */
            if (backtracks < p->backtrack_attempts && m > 1) {              /* :76 */
                int t = 0;
                do {                     /* :78-88 */
                    m--;
                    sum_all -= edge_delta(g, s->p_edges[m]);
                    uint32_t back = s->p_nodes[m];
                    for (uint64_t e = g->row_ptr[back]; e < g->row_ptr[back + 1]; e++)
                        if (!s->visited[g->c_nbr[e]]) { t = 1; break; }
                } while (m != 0 && !t);
                if (m == 0) break;       /* :90-95 */
                ++backtracks;
                n = s->p_nodes[m];       /* :98 */
                continue;
            }
            break;                       /* :101 */
        } else {
            double random = mt_canonical(gen) * sum;                         /* :104 */
            st->draws++;
            sum = 0;
            for (uint64_t c = 0; c < ncand; c++) {                           /* :107-114 */
                sum += (double) metric_of(g, metric, s->cand[c]);
                if (sum >= random) { edge = (int64_t) s->cand[c]; break; }
            }
            /* No prefix reached `random` (NaN / negative metrics): the reference dereferences a
             * null Edge* here (undefined behaviour); the restatement abandons the walk. */
            if (edge < 0) break;
        }
        n = g->c_nbr[edge];              /* :118 */
        s->p_edges[m] = (uint64_t) edge; /* :122 */
        sum_all += edge_delta(g, (uint64_t) edge);
        m++;
        s->p_nodes[m] = n;               /* :121 */
        mark(s, n);                      /* :123 */
        len = path_length(g, s, m, sum_all);   /* :125 */
        if (len <= 0) break;             /* :126-128 */
        if (g->node_anchor[n]) { acceptable = 1; break; }                    /* :131-134 */
        if (len >= p->len_threshold) break;                                  /* :137-142 */
    }
    st->n_nodes = m + 1;
    st->length = path_length(g, s, m, sum_all);
    res->status = 0;
    if (acceptable) {                    /* :146-155 */
        len = path_length(g, s, m, sum_all);
        if (len < 0) res->status = 2;
        else { res->status = 1; res->length = len; store_path(g, s, m, res); }
    }
    st->status = res->status;
    clear_visited(s);
}

/* Rank key of a metric for the deterministic builder's sort: for every pair of non-NaN values, key(a) > key(b)
 * exactly when a > b (with -0 == +0), i.e. the comparator of PathManager.cpp:293-295.  With a NaN in the list (the
 * square root of a negative score) that comparator is not a strict weak ordering and std::stable_sort's result is
 * undefined; oracle and kernel define it: NaNs are ordered by their bit pattern, negative ones below every number,
 * positive ones above (the IEEE total order).  DESIGN.md section 5. */
static inline uint32_t rank_key(float m) {
    uint32_t b;
    memcpy(&b, &m, 4);
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* stable merge sort of edge indices, descending by metric (PathManager.cpp:289-296) */
static void sort_desc_stable(const tlo_graph *g, int metric, uint64_t *a, uint64_t *tmp, uint64_t n) {
    for (uint64_t w = 1; w < n; w *= 2) {
        for (uint64_t lo = 0; lo < n; lo += 2 * w) {
            uint64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                /* take from the right run only if it is strictly greater: stability */
                if (rank_key(metric_of(g, metric, a[j])) > rank_key(metric_of(g, metric, a[i]))) tmp[k++] = a[j++];
                else tmp[k++] = a[i++];
            }
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, n * 8);
    }
}

/* One deterministic walk from (start, first_edge) (PathManager.cpp:210-384).
 * Returns 1 if the caller must stop iterating this anchor's first edges (:224-229). */
static int det_walk(scratch *s, int metric, uint32_t start, uint64_t first_edge, unit_result *res, tlo_unit_stat *st) {
    const tlo_graph *g = s->g;
    memset(st, 0, sizeof(*st));
    res->status = 0;
    uint32_t m = 0;
    int64_t sum_all = 0;
    mark(s, start);                      /* :215 */
    s->p_nodes[0] = start;
    s->p_edges[0] = first_edge;          /* :217 */
    sum_all += edge_delta(g, first_edge);
    m = 1;
    uint32_t node = g->c_nbr[first_edge];   /* :219 */
    mark(s, node);                       /* :221 */
    s->p_nodes[1] = node;
    if (g->node_anchor[node]) {          /* :224-229: appended WITHOUT the length check */
        res->status = 1;
        res->length = path_length(g, s, m, sum_all);
        store_path(g, s, m, res);
        st->status = 1; st->n_nodes = 2; st->length = res->length;
        clear_visited(s);
        return 1;
    }
    int all_ok = 0;
    int step_index = 0;
    int skip_n_best = 0;
    int went_back = 0;
    for (;;) {                           /* :243 */
        uint64_t r0 = g->row_ptr[node], deg = g->row_ptr[node + 1] - r0;
        int go_back = 0;
        uint64_t chosen = 0;
        if (deg == 0) go_back = 1;                                   /* :245 */
        else if ((uint64_t) skip_n_best >= deg - 1) go_back = 1;     /* :267 */
        else {
            for (uint64_t i = 0; i < deg; i++) s->cand[i] = r0 + i; /* :289 */
            sort_desc_stable(g, metric, s->cand, s->tmp, deg);       /* :290-296 */
            st->evals += deg;
            st->steps++;
            int found = 0;
            uint64_t k = (uint64_t) skip_n_best;                     /* :301 */
            while (k < deg - 1) {                                    /* :302-315 */
                chosen = s->cand[k];
                if (!s->visited[g->c_nbr[chosen]]) { found = 1; break; }
                k++;
            }
            if (!found) go_back = 1;                                 /* :318 */
        }
        if (go_back) {
            if (step_index == 0) break;                              /* :246-248,268-271,319-322 */
            if (went_back) break;                                    /* :251-253,274-276,325-327 */
            went_back = 1;
            step_index -= 1;
            skip_n_best += 1;
            m--;                                                     /* pop node + edge */
            sum_all -= edge_delta(g, s->p_edges[m]);
            node = s->p_nodes[m];
            s->visited[node] = 0;                                    /* :262,284,335 (stays in `touched`) */
            continue;
        }
        /* :340-347: push the FIRST entry in list order with the same (q_index, t_index) */
        uint64_t pushed = chosen;
        for (uint64_t e = r0; e < r0 + deg; e++)
            if (g->c_nbr[e] == g->c_nbr[chosen]) { pushed = e; break; }
        s->p_edges[m] = pushed;
        sum_all += edge_delta(g, pushed);
        m++;
        if (!s->visited[node]) { s->visited[node] = 1; s->touched[s->n_touched++] = node; }   /* :355 */
        node = g->c_nbr[chosen];                                     /* :356 */
        s->p_nodes[m] = node;
        if (g->node_anchor[node]) { all_ok = 1; break; }             /* :364-367 */
        step_index += 1;
    }
    st->n_nodes = m + 1;
    st->length = path_length(g, s, m, sum_all);
    if (all_ok) {                        /* :372-384 */
        int64_t len = path_length(g, s, m, sum_all);
        if (len < 0) res->status = 2;
        else { res->status = 1; res->length = len; store_path(g, s, m, res); }
    }
    st->status = res->status;
    clear_visited(s);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* unit drivers */

typedef struct {
    tlo_graph *g;
    int is_mc, metric, seed_mode;
    tlo_walk_params wp;
    uint64_t begin, end;
    unit_result *res;
    tlo_unit_stat *stats;
    uint64_t next;                /* atomic cursor */
    pthread_mutex_t mu;
} job;

static void det_unit_lookup(const tlo_graph *g, uint64_t unit, uint32_t *a_rank) {
    uint32_t lo = 0, hi = g->n_anchors;      /* largest k with det_off[k] <= unit */
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) / 2;
        if (g->det_off[mid] <= unit) lo = mid; else hi = mid;
    }
    *a_rank = lo;
}

static void run_unit(job *j, scratch *s, mt19937 *gen, uint64_t unit) {
    tlo_graph *g = j->g;
    unit_result *res = &j->res[unit - j->begin];
    tlo_unit_stat *st = &j->stats[unit - j->begin];
    if (j->is_mc) {
        uint32_t a_rank = (uint32_t) (unit / (uint64_t) j->wp.rebuild_attempts);
        uint32_t r = (uint32_t) (unit % (uint64_t) j->wp.rebuild_attempts);
        uint32_t start = g->anchors[a_rank];
        if (j->seed_mode == TLO_SEED_PER_WALK) mt_seed(gen, tlo_walk_seed(j->metric, start, r));
        mc_walk(s, j->metric, &j->wp, start, gen, res, st);
    } else {
        uint32_t a_rank;
        det_unit_lookup(g, unit, &a_rank);
        uint32_t start = g->anchors[a_rank];
        uint64_t slot = unit - g->det_off[a_rank];
        det_walk(s, j->metric, start, g->row_ptr[start] + slot, res, st);
    }
}

static void *worker(void *arg) {
    job *j = (job *) arg;
    scratch s;
    scratch_init(&s, j->g);
    mt19937 gen;
    mt_seed(&gen, 5489u);
    for (;;) {
        pthread_mutex_lock(&j->mu);
        uint64_t b = j->next;
        uint64_t e = b + 16 < j->end ? b + 16 : j->end;
        j->next = e;
        pthread_mutex_unlock(&j->mu);
        if (b >= j->end) break;
        for (uint64_t u = b; u < e; u++) run_unit(j, &s, &gen, u);
    }
    scratch_free(&s);
    return NULL;
}

static tlo_paths *collect(job *j, tlo_counters *ctr, tlo_unit_stat *stats_out) {
    uint64_t n_units = j->end - j->begin;
    tlo_counters c = {0, 0, 0, 0, 0};
    uint64_t n_paths = 0, n_nodes = 0;
    for (uint64_t u = 0; u < n_units; u++) {
        c.units++;
        c.steps += j->stats[u].steps;
        c.evals += j->stats[u].evals;
        c.draws += j->stats[u].draws;
        if (j->res[u].status == 1) { n_paths++; n_nodes += j->res[u].n_nodes; }
    }
    c.accepted = n_paths;
    tlo_paths *p = (tlo_paths *) calloc(1, sizeof(tlo_paths));
    p->n_paths = n_paths;
    p->unit = (uint64_t *) malloc((n_paths + 1) * 8);
    p->length = (int64_t *) malloc((n_paths + 1) * 8);
    p->off = (uint64_t *) malloc((n_paths + 1) * 8);
    p->nodes = (uint32_t *) malloc((n_nodes + 1) * 4);
    p->slots = (uint32_t *) malloc((n_nodes + 1) * 4);
    uint64_t k = 0, o = 0;
    for (uint64_t u = 0; u < n_units; u++) {
        unit_result *r = &j->res[u];
        if (r->status != 1) continue;
        p->unit[k] = j->begin + u;
        p->length[k] = r->length;
        p->off[k] = o;
        memcpy(p->nodes + o, r->nodes, (size_t) r->n_nodes * 4);
        memcpy(p->slots + o, r->slots, (size_t) r->n_nodes * 4);
        o += r->n_nodes;
        k++;
        free(r->nodes);
        free(r->slots);
    }
    p->off[k] = o;
    if (ctr) *ctr = c;
    if (stats_out) memcpy(stats_out, j->stats, n_units * sizeof(tlo_unit_stat));
    return p;
}

static tlo_paths *run_job(job *j, int threads, tlo_counters *ctr, tlo_unit_stat *stats_out) {
    uint64_t n_units = j->end - j->begin;
    j->res = (unit_result *) calloc(n_units ? n_units : 1, sizeof(unit_result));
    j->stats = (tlo_unit_stat *) calloc(n_units ? n_units : 1, sizeof(tlo_unit_stat));
    pthread_mutex_init(&j->mu, NULL);
    if (j->is_mc && j->seed_mode == TLO_SEED_SHARED) {
        /* one default-seeded stream per builder call, consumed walk after walk (PathManager.cpp:13-14):
         * units before `begin` must still be walked to advance the stream. */
        scratch s;
        scratch_init(&s, j->g);
        mt19937 gen;
        mt_seed(&gen, 5489u);
        unit_result dummy_r;
        tlo_unit_stat dummy_s;
        for (uint64_t u = 0; u < j->end; u++) {
            if (u < j->begin) {
                uint32_t a_rank = (uint32_t) (u / (uint64_t) j->wp.rebuild_attempts);
                memset(&dummy_r, 0, sizeof(dummy_r));
                mc_walk(&s, j->metric, &j->wp, j->g->anchors[a_rank], &gen, &dummy_r, &dummy_s);
                if (dummy_r.status == 1) { free(dummy_r.nodes); free(dummy_r.slots); }
            } else {
                run_unit(j, &s, &gen, u);
            }
        }
        scratch_free(&s);
    } else if (threads <= 1) {
        j->next = j->begin;
        worker(j);
    } else {
        j->next = j->begin;
        pthread_t *th = (pthread_t *) malloc((size_t) threads * sizeof(pthread_t));
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, j);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    tlo_paths *p = collect(j, ctr, stats_out);
    free(j->res);
    free(j->stats);
    pthread_mutex_destroy(&j->mu);
    return p;
}

tlo_paths *tlo_walk_mc(tlo_graph *g, int metric, const tlo_walk_params *p, int seed_mode, uint64_t unit_begin,
                       uint64_t unit_end, int threads, tlo_counters *ctr, tlo_unit_stat *stats) {
    if (finalize(g)) return NULL;
    uint64_t total = tlo_mc_units(g, p);
    if (unit_end > total) unit_end = total;
    if (unit_begin > unit_end) unit_begin = unit_end;
    if (seed_mode == TLO_SEED_SHARED && threads > 1) return NULL;
    job j;
    memset(&j, 0, sizeof(j));
    j.g = g; j.is_mc = 1; j.metric = metric; j.seed_mode = seed_mode; j.wp = *p; j.begin = unit_begin; j.end = unit_end;
    return run_job(&j, threads, ctr, stats);
}

tlo_paths *tlo_walk_det(tlo_graph *g, int metric, uint64_t unit_begin, uint64_t unit_end, int threads,
                        tlo_counters *ctr, tlo_unit_stat *stats) {
    if (finalize(g)) return NULL;
    uint64_t total = tlo_det_units(g);
    if (unit_end > total) unit_end = total;
    if (unit_begin > unit_end) unit_begin = unit_end;
    job j;
    memset(&j, 0, sizeof(j));
    j.g = g; j.is_mc = 0; j.metric = metric; j.begin = unit_begin; j.end = unit_end;
    return run_job(&j, threads, ctr, stats);
}

void tlo_paths_free(tlo_paths *p) {
    if (!p) return;
    free(p->unit); free(p->length); free(p->off); free(p->nodes); free(p->slots);
    free(p);
}
