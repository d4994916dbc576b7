/* octree.c — exact octree nearest neighbour on the CPU (3-D only): an additional CPU comparator.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see v0_oracle.c): never linked into the product.
 *
 * What it mirrors: the reference's V12 (core.cu:1454-1659), a CPU octree for k == 3 with OpenMP over
 * the queries: cubic cells split at their centre into eight octants down to depth 9, every node keeping
 * the list of its points.  V12 as written indexes points with r_points[*i] where r_points[*i * k] is
 * meant (core.cu:1534, 1548, 1615) and so does not return V0's answers.  This is an
 * independent implementation of the same idea with that indexing done right and V0's exact semantics
 * bolted on, so that its answers can be compared bit for bit with the oracle:
 *   - points are ordered along a 30-bit Morton curve (10 levels of octants of the cloud's bounding cube:
 *     an octree cell is a contiguous range of that order — no per-node point lists);
 *   - every node stores the tight bounding box of its points; a subtree is pruned only if its box
 *     distance, shrunk by a margin that covers V0's own fp32 rounding, still exceeds the best distance
 *     found — a ref that V0 would pick is never skipped;
 *   - leaf distances are V0's fp32 arithmetic (diff, mul, add, t ascending, no FMA) and the winner is
 *     the lexicographic minimum of (V0 distance, index): V0's "first minimum".
 * NaN / INF coordinates are not supported here (the oracle proper is v0_oracle.c).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OC_LEAF 16
#define OC_LEVELS 10

typedef struct {
    int lo, hi;          /* range of the Morton order */
    int first_child;     /* index of the first child node, -1 = leaf */
    int nchild;
    float bmin[3], bmax[3];
} oc_node;

typedef struct {
    int n;
    const float *r;
    uint32_t *code;      /* Morton code of perm[i] */
    int *perm;           /* point indices in Morton order (ties: ascending index) */
    oc_node *nodes;
    int nnodes, cap;
} octree;

static uint32_t spread3(uint32_t v)   /* 10 bits -> every third bit */
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

/* LSD radix sort of (code, index) pairs by code, 3 passes of 10 bits; stable, so equal codes keep
 * ascending point index */
static int sort_by_code(octree *T)
{
    const int n = T->n;
    uint32_t *c2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    int *p2 = (int *)malloc(sizeof(int) * (size_t)n);
    if (!c2 || !p2) {
        free(c2);
        free(p2);
        return 1;
    }
    for (int pass = 0; pass < 3; ++pass) {
        size_t cnt[1025];
        memset(cnt, 0, sizeof(cnt));
        const int sh = 10 * pass;
        for (int i = 0; i < n; ++i) ++cnt[((T->code[i] >> sh) & 1023u) + 1];
        for (int b = 0; b < 1024; ++b) cnt[b + 1] += cnt[b];
        for (int i = 0; i < n; ++i) {
            const size_t d = cnt[(T->code[i] >> sh) & 1023u]++;
            c2[d] = T->code[i];
            p2[d] = T->perm[i];
        }
        uint32_t *tc = T->code;
        T->code = c2;
        c2 = tc;
        int *tp = T->perm;
        T->perm = p2;
        p2 = tp;
    }
    free(c2);
    free(p2);
    return 0;
}

static int new_nodes(octree *T, int count)
{
    if (T->nnodes + count > T->cap) {
        int cap = T->cap * 2 + count;
        oc_node *nn = (oc_node *)realloc(T->nodes, sizeof(oc_node) * (size_t)cap);
        if (!nn) return -1;
        T->nodes = nn;
        T->cap = cap;
    }
    const int first = T->nnodes;
    T->nnodes += count;
    return first;
}

/* node `id` covers perm[lo, hi) whose codes agree above bit 3 * (OC_LEVELS - level) */
static int build(octree *T, int id, int lo, int hi, int level)
{
    T->nodes[id].lo = lo;
    T->nodes[id].hi = hi;
    T->nodes[id].first_child = -1;
    T->nodes[id].nchild = 0;
    for (int d = 0; d < 3; ++d) {
        T->nodes[id].bmin[d] = INFINITY;
        T->nodes[id].bmax[d] = -INFINITY;
    }
    if (hi - lo <= OC_LEAF || level == OC_LEVELS) {
        for (int i = lo; i < hi; ++i)
            for (int d = 0; d < 3; ++d) {
                const float v = T->r[(size_t)T->perm[i] * 3 + d];
                if (v < T->nodes[id].bmin[d]) T->nodes[id].bmin[d] = v;
                if (v > T->nodes[id].bmax[d]) T->nodes[id].bmax[d] = v;
            }
        return 0;
    }
    /* the eight octants are contiguous sub-ranges: find their boundaries */
    const int sh = 3 * (OC_LEVELS - 1 - level);
    int start[9], nch = 0, cs[8], ce[8];
    start[0] = lo;
    int i = lo;
    for (int o = 0; o < 8; ++o) {
        while (i < hi && (int)((T->code[i] >> sh) & 7u) == o) ++i;
        start[o + 1] = i;
        if (start[o + 1] > start[o]) {
            cs[nch] = start[o];
            ce[nch] = start[o + 1];
            ++nch;
        }
    }
    const int first = new_nodes(T, nch);
    if (first < 0) return 1;
    T->nodes[id].first_child = first;
    T->nodes[id].nchild = nch;
    for (int c = 0; c < nch; ++c) {
        if (build(T, first + c, cs[c], ce[c], level + 1)) return 1;
        for (int d = 0; d < 3; ++d) {   /* (T->nodes may have moved: index afresh) */
            if (T->nodes[first + c].bmin[d] < T->nodes[id].bmin[d]) T->nodes[id].bmin[d] = T->nodes[first + c].bmin[d];
            if (T->nodes[first + c].bmax[d] > T->nodes[id].bmax[d]) T->nodes[id].bmax[d] = T->nodes[first + c].bmax[d];
        }
    }
    return 0;
}

static float v0_dist3(const float *a, const float *b)
{
    float s = 0.0f;
    for (int t = 0; t < 3; ++t) {
        const float d = a[t] - b[t];
        s += d * d;
    }
    return s;
}

static double box_dist2(const oc_node *nd, const float *q)
{
    double s = 0.0;
    for (int d = 0; d < 3; ++d) {
        double e = 0.0;
        if (q[d] < nd->bmin[d]) e = (double)nd->bmin[d] - (double)q[d];
        else if (q[d] > nd->bmax[d]) e = (double)q[d] - (double)nd->bmax[d];
        s += e * e;
    }
    return s;
}

typedef struct {
    float best;
    int idx;
} oc_best;

static void search(const octree *T, int id, const float *q, double shrink, oc_best *b)
{
    const oc_node *nd = &T->nodes[id];
    if (nd->first_child < 0) {
        for (int i = nd->lo; i < nd->hi; ++i) {
            const int j = T->perm[i];
            const float d = v0_dist3(q, T->r + (size_t)j * 3);
            if (d < b->best || (d == b->best && j < b->idx)) {
                b->best = d;
                b->idx = j;
            }
        }
        return;
    }
    /* children nearest first */
    int order[8];
    double bd[8];
    const int nch = nd->nchild;
    for (int c = 0; c < nch; ++c) {
        const double v = box_dist2(&T->nodes[nd->first_child + c], q);
        int p = c;
        while (p > 0 && bd[p - 1] > v) {
            bd[p] = bd[p - 1];
            order[p] = order[p - 1];
            --p;
        }
        bd[p] = v;
        order[p] = c;
    }
    for (int c = 0; c < nch; ++c) {
        /* every point of the box is at least sqrt(bd) away; V0's fp32 distance may sit a relative
         * (k + 2) u below the exact one, and an equal distance with a lower index would still win:
         * prune on strict '>' of the shrunk bound only (children are sorted: the rest is farther) */
        if (bd[c] * shrink > (double)b->best) break;
        search(T, nd->first_child + order[c], q, shrink, b);
    }
}

/* indices (and V0 distances) of the nearest ref of each query, k must be 3; 0 on success */
int octree_search(int k, int m, int n, const float *q, const float *r, int *idx, float *dist, int threads)
{
    if (k != 3 || m < 0 || n <= 0) return 1;
    octree T;
    memset(&T, 0, sizeof(T));
    T.n = n;
    T.r = r;
    T.code = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n);
    T.perm = (int *)malloc(sizeof(int) * (size_t)n);
    T.cap = n / 4 + 64;
    T.nodes = (oc_node *)malloc(sizeof(oc_node) * (size_t)T.cap);
    int rc = (!T.code || !T.perm || !T.nodes) ? 2 : 0;
    if (!rc) {
        /* bounding cube of the cloud -> 1024^3 grid */
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) {
                const float v = r[(size_t)i * 3 + d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
        double ext = 0.0;
        for (int d = 0; d < 3; ++d)
            if ((double)hi[d] - (double)lo[d] > ext) ext = (double)hi[d] - (double)lo[d];
        const double scale = ext > 0.0 ? 1023.999 / ext : 0.0;
        for (int i = 0; i < n; ++i) {
            uint32_t g[3];
            for (int d = 0; d < 3; ++d) {
                double t = ((double)r[(size_t)i * 3 + d] - (double)lo[d]) * scale;
                if (!(t >= 0.0)) t = 0.0;
                if (t > 1023.0) t = 1023.0;
                g[d] = (uint32_t)t;
            }
            T.code[i] = spread3(g[0]) | (spread3(g[1]) << 1) | (spread3(g[2]) << 2);
            T.perm[i] = i;
        }
        rc = sort_by_code(&T);
        if (!rc) {
            T.nnodes = 1;
            rc = build(&T, 0, 0, n, 0) ? 2 : 0;
        }
    }
    if (!rc) {
        const double shrink = 1.0 - 4.0 * (3 + 2) * 5.9604644775390625e-08;
#ifdef _OPENMP
        if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
        for (int i = 0; i < m; ++i) {
            oc_best b;
            b.best = INFINITY;
            b.idx = 0;
            search(&T, 0, q + (size_t)i * 3, shrink, &b);
            idx[i] = b.idx;
            if (dist) dist[i] = b.best;
        }
    }
    free(T.code);
    free(T.perm);
    free(T.nodes);
    (void)threads;
    return rc;
}
