// libsqgr internal: shuffled cluster-label vectors for permutation tests that are not tied to a graph.
// Implemented in sqgr_nhood.hip on top of the neighbourhood-enrichment label generators.
#pragma once
#include "sqgr_common.h"

namespace sqgr {

struct LabelShuffler;  // opaque

int label_shuffler_create(sqgr_ctx* ctx, int64_t n, const int32_t* labels, int K, LabelShuffler** out);
void label_shuffler_destroy(LabelShuffler* s);

// device generator (Philox-keyed Feistel bijections): slab[(q*n + i)*32 + b] = label of item i in permutation
// perm0 + q*32 + b, for q < nb.  keys_ws: nb*32*8 words of scratch.
int label_shuffler_philox(LabelShuffler* s, uint64_t seed, int64_t perm0, int nb, uint32_t* keys_ws, uint8_t* slab, hipStream_t st);

// numpy streams (PCG64 + Generator.shuffle): W[i*stride + q] = label of item i in the permutation generator q yields,
// q < pc.  states_dev: pc rows [state_hi, state_lo, inc_hi, inc_lo] on the device.
int label_shuffler_pcg64(LabelShuffler* s, const uint64_t* states_dev, int64_t pc, int64_t stride, uint8_t* W, hipStream_t st);

// More than 256 labels (K <= 65535 without a graph): 16-bit label rows of 16 permutations, slab16[(q*n + i)*16 + b] = label of item i in
// permutation perm0 + q*16 + b (device generator; keys_ws: nb * label_shuffler_key_words16() words) or in the permutation
// generator q*16 + b yields (numpy streams; labels of generators >= pc are 0).
bool label_shuffler_wide(const LabelShuffler* s);
size_t label_shuffler_key_words16();
int label_shuffler_philox16(LabelShuffler* s, uint64_t seed, int64_t perm0, int nb, uint32_t* keys_ws, uint16_t* slab16, hipStream_t st);
int label_shuffler_pcg64_16(LabelShuffler* s, const uint64_t* states_dev, int64_t pc, uint16_t* slab16, hipStream_t st);

}  // namespace sqgr
