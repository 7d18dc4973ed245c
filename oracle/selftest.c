/* selftest.c -- sanitizer driver for the oracle (built with -fsanitize=address,undefined by `make asan-test`).
 * TEST INFRASTRUCTURE ONLY.  Builds a small closed room, the three partitions, shoots a few thousand rays
 * (inside, outside, degenerate) through every entry point and cross-checks voxel == brute force. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hare_oracle.h"

static uint64_t sm(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double u01(uint64_t *s) { return (double)(sm(s) >> 11) * (1.0 / 9007199254740992.0); }

int main(void)
{
    /* box 4 x 3 x 2 split into n x n quads per face, each quad two triangles (+ a few real quads) */
    const int n = 6;
    const double L[3] = {4.0, 3.0, 2.0};
    int P = 6 * n * n * 2, p = 0;
    double *verts = calloc((size_t)P * 12, sizeof(double));
    int32_t *nverts = calloc((size_t)P, sizeof(int32_t));
    for (int f = 0; f < 6; ++f) {
        int a = f / 2, b = (a + 1) % 3, c = (a + 2) % 3;
        double w = (f & 1) ? L[a] : 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double q[4][3];
                for (int k = 0; k < 4; ++k) {
                    int di = (k == 1 || k == 2), dj = (k >= 2);
                    q[k][a] = w;
                    q[k][b] = L[b] * (i + di) / n;
                    q[k][c] = L[c] * (j + dj) / n;
                }
                if ((i + j) % 5 == 0) { /* keep as one quadrilateral + a degenerate filler triangle far away */
                    memcpy(&verts[(size_t)p * 12], q, sizeof q);
                    nverts[p++] = 4;
                    double t[3][3] = {{9, 9, 9}, {9.5, 9, 9}, {9, 9.5, 9}};
                    memcpy(&verts[(size_t)p * 12], t, sizeof t);
                    nverts[p++] = 3;
                } else {
                    double t0[3][3], t1[3][3];
                    memcpy(t0[0], q[0], 24); memcpy(t0[1], q[1], 24); memcpy(t0[2], q[2], 24);
                    memcpy(t1[0], q[0], 24); memcpy(t1[1], q[2], 24); memcpy(t1[2], q[3], 24);
                    memcpy(&verts[(size_t)p * 12], t0, sizeof t0); nverts[p++] = 3;
                    memcpy(&verts[(size_t)p * 12], t1, sizeof t1); nverts[p++] = 3;
                }
            }
    }
    double *ing = calloc((size_t)P * 12, sizeof(double));
    int32_t nv = ho_build_topology(verts, nverts, P, ing);
    double *normals = calloc((size_t)P * 3, sizeof(double));
    ho_polygon_normals(ing, nverts, P, normals);
    ho_topology T;
    memset(&T, 0, sizeof T);
    T.P = P; T.verts = ing; T.nverts = nverts; T.normals = normals;
    ho_finish_topology_bounds(ing, nverts, P, T.min, T.max);

    ho_voxel_grid *g0 = ho_voxel_build(&T, 1, 5, 0), *g1 = ho_voxel_build(&T, 1, 5, 1);
    ho_voxel_grid *ga = ho_voxel_build_adaptive(&T, 1, 4, 6);
    size_t nc = 125;
    if (memcmp(ho_voxel_cell_start(g0, 0), ho_voxel_cell_start(g1, 0), (nc + 1) * 4) != 0) { puts("list mismatch"); return 1; }
    ho_octree *oc = ho_octree_build(&T, 1, 4, 6);
    ho_kdtree *kd = ho_kdtree_build(&T, 1, 6, 6);

    const int N = 4000;
    ho_ray *rays = calloc(N, sizeof(ho_ray));
    uint64_t s = 42;
    for (int i = 0; i < N; ++i) {
        double o[3], d[3];
        for (int a = 0; a < 3; ++a) { o[a] = (u01(&s) * 1.6 - 0.3) * L[a]; d[a] = u01(&s) * 2 - 1; }
        if (i % 97 == 0) d[0] = 0.0;
        if (i % 101 == 0) d[1] = -0.0;
        if (i % 211 == 0) o[2] = NAN;
        rays[i] = (ho_ray){o[0], o[1], o[2], d[0], d[1], d[2]};
    }
    ho_xevent *ev = calloc(N, sizeof(ho_xevent)), *e2 = calloc(N, sizeof(ho_xevent));
    ho_counters c;
    ho_voxel_shoot_batch(g1, &T, 0, N, rays, NULL, NULL, 1, 1, 3, ev, &c);
    ho_voxel_shoot_batch(ga, &T, 0, N, rays, NULL, NULL, 1, 1, 1, e2, &c);
    ho_octree_shoot_batch(oc, &T, 0, N, rays, NULL, NULL, 2, e2, &c);
    ho_kdtree_shoot_batch(kd, &T, 0, N, rays, NULL, NULL, 1, 2, e2, &c);
    int bad = 0, hits = 0;
    for (int i = 0; i < N; ++i) {
        ho_xevent b;
        ho_brute_shoot(&T, &rays[i], -1, -1, 0, &b);
        hits += ev[i].hit;
        int inside = rays[i].x > 0 && rays[i].x < L[0] && rays[i].y > 0 && rays[i].y < L[1] && rays[i].z > 0 && rays[i].z < L[2];
        if (inside && ev[i].hit && b.t != ev[i].t) bad++;
        if (ev[i].hit) { ho_ray r2; ho_reflect(&T, &rays[i], &ev[i], &r2); (void)r2; }
    }
    ho_voxel_pool *pool = ho_voxel_pool_new(g1, &T);
    for (int i = 0; i < 600; ++i) { ho_ray r = rays[i]; ho_voxel_pool_shoot(pool, &r, i, 0, -1, -1, &e2[i]); }
    ho_voxel_pool_free(pool);
    printf("selftest: %d polys (%d vertices), %d rays, %d hits, %d mismatches\n", P, nv, N, hits, bad);
    ho_voxel_free(g0); ho_voxel_free(g1); ho_voxel_free(ga); ho_octree_free(oc); ho_kdtree_free(kd);
    free(verts); free(nverts); free(ing); free(normals); free(rays); free(ev); free(e2);
    return bad != 0;
}
