// wrap_shim.cpp — TEST INFRASTRUCTURE: links integration/faucet_binding.cpp into the COMPILED reference without touching its sources.
//
// A maintainer would edit two call sites of src/Faucet.cpp.  To prove that the binding really is a drop-in for those call
// sites -- and that the reference's own downstream stage runs on what the C ABI hands back -- oracle/Makefile (targets ref_gpu / ref_stub)
// links the reference's unmodified objects with `-Wl,--wrap=` on three symbols, and the __wrap_ functions below route them to the binding:
//     load_two_filters(Bloom*, Bloom*, string, bool, bool)      src/Faucet.cpp:220     -> gpu_load_two_filters
//     ReadScanner::scanReads(bool, bool, bool)                   src/Faucet.cpp:244     -> gpu_scan / gpu_scan_paired
//     ReadScanner::printScanSummary()                            src/Faucet.cpp:245     -> nothing (the binding prints those lines itself)
// Everything else of the reference (argument handling, sizing, Bloom::dump, JunctionMap::writeToFile, the pair filters' dump, buildContigGraph
// and all of Stage 3) runs as compiled from /root/reference.  Nothing of this file enters the product.
// every standard header the reference's headers pull in comes first, so that the access trick below touches the reference's classes only
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <iterator>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <new>
#include <queue>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#define private public          // the scanner's members (junctionMap, the pair filters, reads_file) are what scanReads works on
#include "../src/ReadScanner.h"
#undef private

extern bool no_cleaning, paired_ends;      // src/Faucet.h:46-47, defined in Faucet.o

void gpu_configure(bool cleaning, bool paired_ends, int n_gpus);
void gpu_load_two_filters(Bloom* bloo1, Bloom* bloo2, std::string reads_filename, bool fastq, bool mercy);
void gpu_scan(JunctionMap* junctionMap, std::string read_scan_file, bool fastq, Bloom* short_pair_filter);
void gpu_scan_paired(JunctionMap* junctionMap, std::string read_scan_file, bool fastq, Bloom* short_pair_filter, Bloom* long_pair_filter,
                     bool no_cleaning);

extern "C" {

void __wrap__Z16load_two_filtersP5BloomS0_NSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEEbb(Bloom* bloo1, Bloom* bloo2, std::string reads_filename,
                                                                                                  bool fastq, bool mercy) {
    gpu_configure(!no_cleaning, paired_ends, 0);       // (0: the number of GPUs from $FAUCET_GPUS -- the reference has no flag for it)
    gpu_load_two_filters(bloo1, bloo2, reads_filename, fastq, mercy);
}

void __wrap__ZN11ReadScanner9scanReadsEbbb(ReadScanner* self, bool fastq, bool paired, bool no_clean) {
    if (paired) gpu_scan_paired(self->junctionMap, self->reads_file, fastq, no_clean ? NULL : self->short_pair_filter, self->long_pair_filter, no_clean);
    else gpu_scan(self->junctionMap, self->reads_file, fastq, no_clean ? NULL : self->short_pair_filter);
}

void __wrap__ZN11ReadScanner16printScanSummaryEv(ReadScanner*) {}

}  // extern "C"
