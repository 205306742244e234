/*
 * ptx_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, single
 * thread) of the index-producing steps of ProxyTransformation's preshape hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (proxytransformation_amd) never does.
 *
 * Reference alias: PRE = embodiedscan/models/necks/preshape_norm_reverse_drop.py
 *
 * PARITY PIN STATUS
 *   - oracle_fps / oracle_select_clusters / oracle_pt_replace / oracle_remove_points
 *     are pinned by the tests/golden npz fixtures, captured from PRE itself
 *     (FPS through PRE's own in-file statement sample_farthest_points_naive, PRE:527-625,
 *     gathers through PRE's masked_gather, PRE:627-672).
 *   - oracle_ball_query restates pytorch3d.ops.ball_query (pytorch3d is NOT vendored
 *     under /root/reference and is unpinned in requirements/run.txt:6), following the
 *     published algorithm of pytorch3d/csrc/ball_query/ball_query_cpu.cpp: idx
 *     initialised to -1, radius2 = radius*radius in float, for every centre scan the
 *     points in index order, dist2 accumulated over d = 0,1,2 in float, keep while
 *     dist2 < radius2 (strict), stop after K hits.  The golden vectors pin it only
 *     against an independent torch restatement of that same published algorithm
 *     (tests/golden/gen_golden.py), anchored on PRE's call sites (PRE:56, PRE:65) and
 *     output conventions (PRE:94, PRE:372, PRE:478): "parity unpinned" for the
 *     third-party arithmetic itself.
 *
 * Every float expression below is written so that the compiler cannot contract
 * a*b+c into an FMA (build with -ffp-contract=off; see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* PRE:41-48 (DeformablePointCluster.init_uniform_cluster_center).
 * lin: the gs values of torch.linspace(0,1,gs) supplied by the caller (SURVEY H3:
 * torch's two-sided linspace formula is not re-derived).  meshgrid default 'ij':
 * cell m = ix*gs*gs + iy*gs + iz (PRE:44-45).
 * centre = (min + margin) + lin * ((max - min) - 2*margin)   (PRE:48, python
 * operator order). */
EXPORT void oracle_grid_centers(const float *points, int B, int N, const float *lin, int gs,
                                float margin, float *centers /*B,gs^3,3*/,
                                float *mn /*B,3*/, float *mx /*B,3*/)
{
    const float two_margin = 2.0f * margin;
    for (int b = 0; b < B; ++b) {
        const float *p = points + (size_t)b * N * 3;
        float lo[3] = {p[0], p[1], p[2]}, hi[3] = {p[0], p[1], p[2]};
        for (int i = 1; i < N; ++i)
            for (int d = 0; d < 3; ++d) {
                float v = p[(size_t)i * 3 + d];
                if (v < lo[d]) lo[d] = v;
                if (v > hi[d]) hi[d] = v;
            }
        for (int d = 0; d < 3; ++d) { mn[b * 3 + d] = lo[d]; mx[b * 3 + d] = hi[d]; }
        int M = gs * gs * gs;
        for (int m = 0; m < M; ++m) {
            int ijk[3] = {m / (gs * gs), (m / gs) % gs, m % gs};
            for (int d = 0; d < 3; ++d) {
                float base = lo[d] + margin;
                float span = (hi[d] - lo[d]) - two_margin;
                float scaled = lin[ijk[d]] * span;
                centers[((size_t)b * M + m) * 3 + d] = base + scaled;
            }
        }
    }
}

/* pytorch3d.ops.ball_query semantics (call sites PRE:56, PRE:65), see header.
 * idx: (B,M,K) int64 padded -1; cluster: (B,M,K,3) gathered xyz padded 0.0
 * (masked_gather, PRE:627-672).  scanned (optional, may be NULL): number of
 * points examined before the scan stopped (for bench statistics only). */
EXPORT void oracle_ball_query(const float *centers, const float *points, int B, int M, int N, int K,
                              float radius, int64_t *idx, float *cluster, int32_t *scanned)
{
    const float radius2 = radius * radius;
    for (int b = 0; b < B; ++b) {
        const float *p2 = points + (size_t)b * N * 3;
        for (int i = 0; i < M; ++i) {
            const float *c = centers + ((size_t)b * M + i) * 3;
            int64_t *oi = idx + ((size_t)b * M + i) * K;
            float *oc = cluster + ((size_t)b * M + i) * K * 3;
            for (int k = 0; k < K; ++k) { oi[k] = -1; oc[3 * k] = oc[3 * k + 1] = oc[3 * k + 2] = 0.0f; }
            int count = 0, j = 0;
            for (; j < N && count < K; ++j) {
                float dist2 = 0.0f;
                for (int d = 0; d < 3; ++d) {
                    float diff = c[d] - p2[(size_t)j * 3 + d];
                    float sq = diff * diff;
                    dist2 = dist2 + sq;
                }
                if (dist2 < radius2) {
                    oi[count] = j;
                    oc[3 * count + 0] = p2[(size_t)j * 3 + 0];
                    oc[3 * count + 1] = p2[(size_t)j * 3 + 1];
                    oc[3 * count + 2] = p2[(size_t)j * 3 + 2];
                    ++count;
                }
            }
            if (scanned) scanned[(size_t)b * M + i] = j;
        }
    }
}

/* Farthest point sampling, pytorch3d.ops.sample_farthest_points semantics with
 * lengths=None, random_start_point=False (call site PRE:393; in-file statement
 * PRE:527-625): pick[0] = 0; closest[p] = min(closest[p], |x_p - x_last|^2);
 * next = first arg-max of closest (PRE:609, 613). */
EXPORT void oracle_fps(const float *pts /*B,P,3*/, int B, int P, int Kd, int64_t *picks /*B,Kd*/)
{
    float *closest = (float *)malloc(sizeof(float) * (size_t)(P > 0 ? P : 1));
    for (int b = 0; b < B; ++b) {
        const float *x = pts + (size_t)b * P * 3;
        int64_t *out = picks + (size_t)b * Kd;
        for (int k = 0; k < Kd; ++k) out[k] = -1;
        for (int p = 0; p < P; ++p) closest[p] = INFINITY;
        int sel = 0;
        if (Kd > 0) out[0] = 0;
        int kn = Kd < P ? Kd : P;                     /* PRE:595 */
        for (int k = 1; k < kn; ++k) {
            float best = -1.0f; int besti = 0;
            for (int p = 0; p < P; ++p) {
                float d2 = 0.0f;
                for (int d = 0; d < 3; ++d) {
                    float diff = x[(size_t)sel * 3 + d] - x[(size_t)p * 3 + d];
                    float sq = diff * diff;
                    d2 = d2 + sq;
                }
                if (d2 < closest[p]) closest[p] = d2;
                if (closest[p] > best) { best = closest[p]; besti = p; }   /* first max */
            }
            sel = besti;
            out[k] = sel;
        }
    }
    free(closest);
}

/* dynamic_cluster_dropout, PRE:352-420, with the argsort tie-break pinned.
 *   padding_counts = #(idx == -1) per cluster                           (PRE:372)
 *   order          = first Mt of argsort(padding_counts)                (PRE:376-379)
 *                    - production pin: STABLE ascending;
 *                    - if order_override != NULL it is used verbatim (B,Mt): this is how
 *                      the tests replay torch's as-shipped unstable permutation.
 *   picks          = FPS over the re-ordered centres, Kd = Mt - Mk      (PRE:388-393)
 *   keep           = ascending positions not in picks, first Mk         (PRE:395-408)
 * Outputs (all in ORIGINAL cluster ids through `order`):
 *   order_out (B,Mt), picks_out (B,Kd) positions into order, keep_out (B,Mk) positions.
 */
EXPORT void oracle_select_clusters(const int64_t *idx /*B,M,K*/, const float *centers /*B,M,3*/,
                                   int B, int M, int K, int Mt, int Mk,
                                   const int64_t *order_override,
                                   int64_t *pad_counts /*B,M*/, int64_t *order_out /*B,Mt*/,
                                   int64_t *picks_out /*B,Kd*/, int64_t *keep_out /*B,Mk*/)
{
    int Kd = Mt - Mk;
    float *sc = (float *)malloc(sizeof(float) * 3 * (size_t)(Mt > 0 ? Mt : 1));
    char *picked = (char *)malloc((size_t)(Mt > 0 ? Mt : 1));
    for (int b = 0; b < B; ++b) {
        int64_t *pc = pad_counts + (size_t)b * M;
        for (int m = 0; m < M; ++m) {
            int c = 0;
            for (int k = 0; k < K; ++k) c += idx[((size_t)b * M + m) * K + k] == -1;
            pc[m] = c;
        }
        int64_t *ord = order_out + (size_t)b * Mt;
        if (order_override) {
            memcpy(ord, order_override + (size_t)b * Mt, sizeof(int64_t) * (size_t)Mt);
        } else {
            int t = 0;                                 /* stable counting sort, keys 0..K */
            for (int c = 0; c <= K && t < Mt; ++c)
                for (int m = 0; m < M && t < Mt; ++m)
                    if (pc[m] == c) ord[t++] = m;
        }
        for (int t = 0; t < Mt; ++t)
            for (int d = 0; d < 3; ++d)
                sc[t * 3 + d] = centers[((size_t)b * M + ord[t]) * 3 + d];
        int64_t *pk = picks_out + (size_t)b * Kd;
        oracle_fps(sc, 1, Mt, Kd, pk);
        memset(picked, 0, (size_t)Mt);
        for (int k = 0; k < Kd; ++k) if (pk[k] >= 0) picked[pk[k]] = 1;
        int64_t *kp = keep_out + (size_t)b * Mk;
        int n = 0;
        for (int t = 0; t < Mt && n < Mk; ++t) if (!picked[t]) kp[n++] = t;
        for (; n < Mk; ++n) kp[n] = -1;               /* cannot happen: survivors >= Mk */
    }
    free(sc); free(picked);
}

/* pt_replace, PRE:472-498, with the duplicate-target rule pinned to "last writer in
 * flat (m,k) order wins" (what single-threaded index_put_ does).
 * points is modified in place (the reference scatters into its stacked copy). */
EXPORT void oracle_pt_replace(float *points /*B,N,3*/, const int64_t *idx /*B,Mk,K*/,
                              const float *newc /*B,Mk,K,3*/, int B, int N, int Mk, int K)
{
    for (int b = 0; b < B; ++b)
        for (int s = 0; s < Mk * K; ++s) {
            int64_t j = idx[(size_t)b * Mk * K + s];
            if (j == -1) continue;
            for (int d = 0; d < 3; ++d)
                points[((size_t)b * N + j) * 3 + d] = newc[((size_t)b * Mk * K + s) * 3 + d];
        }
}

/* remove_points_by_index, PRE:501-525: drop every point whose index occurs in
 * drop_idx (B,Nd) (-1 matches nothing), keep original order.  out is (B,N,3)
 * capacity; counts[b] = surviving points of scene b. */
EXPORT void oracle_remove_points(const float *points /*B,N,3*/, const int64_t *drop_idx /*B,Nd*/,
                                 int B, int N, int Nd, float *out, int64_t *counts)
{
    char *drop = (char *)malloc((size_t)(N > 0 ? N : 1));
    for (int b = 0; b < B; ++b) {
        memset(drop, 0, (size_t)N);
        for (int i = 0; i < Nd; ++i) {
            int64_t j = drop_idx[(size_t)b * Nd + i];
            if (j >= 0 && j < N) drop[j] = 1;
        }
        int64_t n = 0;
        for (int i = 0; i < N; ++i) {
            if (drop[i]) continue;
            for (int d = 0; d < 3; ++d)
                out[((size_t)b * N + n) * 3 + d] = points[((size_t)b * N + i) * 3 + d];
            ++n;
        }
        counts[b] = n;
    }
    free(drop);
}
