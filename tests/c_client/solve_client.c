/* A host with no Python and no PyTorch in it: the C ABI of include/metrotrpl_amd.h as a cgo / JNI / FFI binding would
 * see it -- plain pointers and sizes.  Reads a batch (descriptors exactly as mtrpl_solve_desc lays them out, the time
 * grid) from a file, calls mtrpl_solve_batch (replaces forward_solver.solve, forward_solver.py:41-203, B calls at once),
 * writes the traces back.  tests/test_gpu_c_client.py builds it with gcc and compares with the Python path bit for bit.
 *
 *   gcc -O2 -I include -o solve_client tests/c_client/solve_client.c -L metrotrpl_amd/lib -lmetrotrpl_amd
 *   solve_client batch.bin out.bin [host_chunk]   with a third argument the output buffer is pinned memory from
 *                                                 mtrpl_host_alloc and the batch runs as sub-launches of host_chunk solves
 *                                                 (the pipelined host path, include/metrotrpl_amd.h): same traces
 *   file in : int64 B, int64 nt, B x mtrpl_solve_desc, nt x double
 *   file out: B x int32 status, B*nt x double                                                       */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "metrotrpl_amd.h"

int main(int argc, char **argv)
{
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s batch.bin out.bin [host_chunk]\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t B = 0, nt = 0;
    if (fread(&B, 8, 1, f) != 1 || fread(&nt, 8, 1, f) != 1 || B <= 0 || nt <= 0) { fprintf(stderr, "bad header\n"); return 2; }
    mtrpl_solve_desc *descs = malloc(sizeof(mtrpl_solve_desc) * (size_t)B);
    double *t = malloc(sizeof(double) * (size_t)nt);
    double *out = NULL;
    int32_t *status = malloc(sizeof(int32_t) * (size_t)B);
    if (fread(descs, sizeof(mtrpl_solve_desc), (size_t)B, f) != (size_t)B || fread(t, 8, (size_t)nt, f) != (size_t)nt) {
        fprintf(stderr, "short file\n");
        return 2;
    }
    fclose(f);
    mtrpl_ctx *ctx = NULL;
    int rc = mtrpl_ctx_create(0, &ctx);
    if (rc != MTRPL_OK) { fprintf(stderr, "mtrpl_ctx_create: %d (no MI355X?)\n", rc); return 3; }
    const int pipelined = argc == 4;
    if (pipelined) {
        void *p = NULL;
        if (mtrpl_host_alloc(ctx, (int64_t)sizeof(double) * B * nt, &p) != MTRPL_OK ||
            mtrpl_set_option(ctx, "host_chunk", atof(argv[3])) != MTRPL_OK || mtrpl_set_option(ctx, "host_streams", 3.0) != MTRPL_OK) {
            fprintf(stderr, "pinned buffer / options: %s\n", mtrpl_last_error(ctx));
            return 6;
        }
        out = p;
    } else {
        out = malloc(sizeof(double) * (size_t)(B * nt));
    }
    rc = mtrpl_solve_batch(ctx, B, descs, NULL, 0, t, nt, out, B * nt, status, NULL);
    if (rc != MTRPL_OK) { fprintf(stderr, "mtrpl_solve_batch: %d: %s\n", rc, mtrpl_last_error(ctx)); return 4; }
    /* the reference's error behaviour at this boundary: a bad argument is an error code and a message, not a crash */
    rc = mtrpl_solve_batch(ctx, B, NULL, NULL, 0, t, nt, out, B * nt, status, NULL);
    if (rc == MTRPL_OK) { fprintf(stderr, "a NULL descriptor array was accepted\n"); return 5; }
    printf("%s: %lld solves, bad-argument call -> %d (%s)\n", mtrpl_version(), (long long)B, rc, mtrpl_last_error(ctx));
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    fwrite(status, sizeof(int32_t), (size_t)B, f);
    fwrite(out, sizeof(double), (size_t)(B * nt), f);
    fclose(f);
    if (pipelined) {
        if (mtrpl_host_free(ctx, out) != MTRPL_OK || mtrpl_host_free(ctx, descs) == MTRPL_OK /* not ours: must be refused */ ||
            mtrpl_ctx_trim(ctx) != MTRPL_OK) {
            fprintf(stderr, "mtrpl_host_free / mtrpl_ctx_trim misbehaved\n");
            return 7;
        }
    } else {
        free(out);
    }
    mtrpl_ctx_destroy(ctx);
    free(descs); free(t); free(status);
    return 0;
}
