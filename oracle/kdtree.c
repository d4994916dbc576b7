/* kdtree.c — exact k-d tree nearest neighbour on the CPU: an additional CPU comparator.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see v0_oracle.c): never linked into the product.
 *
 * What it mirrors: the reference's only other working algorithm family, the CPU k-d tree of
 * V10 (core.cu:1060-1163: build by the dimension of largest spread, median split, recursive
 * search with plane pruning, falls back to V0 above 16 dimensions, core.cu:1148-1149).  This is
 * an independent implementation with V0's exact semantics bolted on, so that its answers can be
 * compared bit for bit with the oracle:
 *   - leaf distances are V0's fp32 arithmetic (diff, mul, add, t ascending, no FMA);
 *   - the winner is the lexicographic minimum of (V0 distance, index): V0's "first minimum";
 *   - a subtree is pruned only if its plane distance, shrunk by a margin that covers V0's own
 *     rounding, still exceeds the best distance — so a ref that V0 would pick is never skipped.
 * NaN / INF coordinates are not supported here (the oracle proper is v0_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KD_LEAF 8

typedef struct {
    int k, n;
    const float *r;
    int *perm;        /* point indices, partitioned in place */
    int *split_dim;   /* per node (heap order over [lo, hi) ranges): -1 = leaf */
    float *split_val;
    int nodes;
} kdtree;

static float v0_dist(const float *a, const float *b, int k)
{
    float s = 0.0f;
    for (int t = 0; t < k; ++t) {
        const float d = a[t] - b[t];
        s += d * d;
    }
    return s;
}

/* quickselect on perm[lo..hi) by coordinate d: element mid ends in sorted position */
static void select_nth(const kdtree *T, int lo, int hi, int mid, int d)
{
    const float *r = T->r;
    const int k = T->k;
    int *p = T->perm;
    while (hi - lo > 1) {
        const float pv = r[(size_t)p[lo + (hi - lo) / 2] * k + d];
        int i = lo, j = hi - 1;
        while (i <= j) {
            while (r[(size_t)p[i] * k + d] < pv) ++i;
            while (r[(size_t)p[j] * k + d] > pv) --j;
            if (i <= j) {
                const int t = p[i];
                p[i] = p[j];
                p[j] = t;
                ++i;
                --j;
            }
        }
        if (mid <= j) hi = j + 1;
        else if (mid >= i) lo = i;
        else return;
    }
}

static void build(kdtree *T, int node, int lo, int hi)
{
    if (node >= T->nodes) return;
    if (hi - lo <= KD_LEAF) {
        T->split_dim[node] = -1;
        return;
    }
    /* dimension of largest extent */
    int best_d = 0;
    float best_e = -1.0f;
    for (int d = 0; d < T->k; ++d) {
        float mn = INFINITY, mx = -INFINITY;
        for (int i = lo; i < hi; ++i) {
            const float v = T->r[(size_t)T->perm[i] * T->k + d];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
        if (mx - mn > best_e) {
            best_e = mx - mn;
            best_d = d;
        }
    }
    const int mid = lo + (hi - lo) / 2;
    select_nth(T, lo, hi, mid, best_d);
    T->split_dim[node] = best_d;
    T->split_val[node] = T->r[(size_t)T->perm[mid] * T->k + best_d];
    build(T, 2 * node + 1, lo, mid);
    build(T, 2 * node + 2, mid, hi);
}

typedef struct {
    float best;
    int idx;
} kd_best;

static void search(const kdtree *T, int node, int lo, int hi, const float *q, double shrink, kd_best *b)
{
    if (node >= T->nodes || T->split_dim[node] < 0) {
        for (int i = lo; i < hi; ++i) {
            const int j = T->perm[i];
            const float d = v0_dist(q, T->r + (size_t)j * T->k, T->k);
            if (d < b->best || (d == b->best && j < b->idx)) {
                b->best = d;
                b->idx = j;
            }
        }
        return;
    }
    const int d = T->split_dim[node];
    const int mid = lo + (hi - lo) / 2;
    const double delta = (double)q[d] - (double)T->split_val[node];
    const int near_left = delta < 0.0;
    if (near_left) search(T, 2 * node + 1, lo, mid, q, shrink, b);
    else search(T, 2 * node + 2, mid, hi, q, shrink, b);
    /* every point of the far side is at least |delta| away in dimension d; V0's fp32 distance may
     * sit a relative (k + 2) u below the exact one, and an equal distance with a lower index
     * would still win: prune on strict '>' of the shrunk bound only */
    if (delta * delta * shrink > (double)b->best) return;
    if (near_left) search(T, 2 * node + 2, mid, hi, q, shrink, b);
    else search(T, 2 * node + 1, lo, mid, q, shrink, b);
}

/* indices (and V0 distances) of the nearest ref of each query; 0 on success */
int kdtree_search(int k, int m, int n, const float *q, const float *r, int *idx, float *dist, int threads)
{
    if (k <= 0 || m < 0 || n <= 0) return 1;
    kdtree T;
    T.k = k;
    T.n = n;
    T.r = r;
    T.perm = (int *)malloc(sizeof(int) * (size_t)n);
    int depth = 0;
    while (((size_t)KD_LEAF << depth) < (size_t)n) ++depth;
    T.nodes = (1 << (depth + 1)) - 1;
    T.split_dim = (int *)malloc(sizeof(int) * (size_t)T.nodes);
    T.split_val = (float *)malloc(sizeof(float) * (size_t)T.nodes);
    if (!T.perm || !T.split_dim || !T.split_val) {
        free(T.perm);
        free(T.split_dim);
        free(T.split_val);
        return 2;
    }
    for (int i = 0; i < n; ++i) T.perm[i] = i;
    for (int i = 0; i < T.nodes; ++i) T.split_dim[i] = -1;
    build(&T, 0, 0, n);
    const double shrink = 1.0 - 4.0 * (k + 2) * 5.9604644775390625e-08;
#ifdef _OPENMP
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
    for (int i = 0; i < m; ++i) {
        kd_best b;
        b.best = INFINITY;
        b.idx = 0;
        search(&T, 0, 0, n, q + (size_t)i * k, shrink, &b);
        idx[i] = b.idx;
        if (dist) dist[i] = b.best;
    }
    free(T.perm);
    free(T.split_dim);
    free(T.split_val);
    (void)threads;
    return 0;
}
