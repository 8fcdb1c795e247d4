/* What a per-key order of the junction walk leaves to run concurrently, on real dependency data (scripts/cross_rank_order.py).
 *
 * Reads are walked by N ranks of W walkers each (a walker = a wave that holds one piece at a time, as in k_walk_ko).  A rank hands its reads
 * out in file order to the walker that is free first; a read may touch a junction k-mer only after the previous read IN FILE ORDER that holds
 * that k-mer has finished (ReadScanner.cpp:61-231: one map, mutated in read order -- the per-key form of that order), and it keeps its walker
 * while it waits.  A read costs one time unit.  Every dependency points to a lower read number, so one sweep in file order settles all times.
 *
 *   occ_start[n_reads + 1], occ_key[]: the junction k-mers (dense ids) on each read
 *   rank_of[n_reads]:                  which rank walks the read
 * returns the makespan; wait_sum = total time reads spent holding a walker without walking.
 */
#include <stdint.h>
#include <stdlib.h>

static void sift_down(double* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        double t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}

double simulate(int64_t n_reads, const int64_t* occ_start, const int32_t* occ_key, int64_t n_keys, const int32_t* rank_of, int n_ranks,
                int walkers, double* wait_sum, double* rank_end) {
    double* key_ready = (double*)calloc((size_t)n_keys, sizeof(double));
    double* heaps = (double*)calloc((size_t)n_ranks * walkers, sizeof(double));
    double makespan = 0, waited = 0;
    for (int64_t i = 0; i < n_reads; i++) {
        double* h = heaps + (size_t)rank_of[i] * walkers;
        const double start = h[0];
        double ready = start;
        for (int64_t o = occ_start[i]; o < occ_start[i + 1]; o++)
            if (key_ready[occ_key[o]] > ready) ready = key_ready[occ_key[o]];
        const double end = ready + 1.0;
        waited += ready - start;
        for (int64_t o = occ_start[i]; o < occ_start[i + 1]; o++) key_ready[occ_key[o]] = end;
        h[0] = end;
        sift_down(h, walkers, 0);
        if (end > makespan) makespan = end;
        if (end > rank_end[rank_of[i]]) rank_end[rank_of[i]] = end;
    }
    *wait_sum = waited;
    free(key_ready);
    free(heaps);
    return makespan;
}
