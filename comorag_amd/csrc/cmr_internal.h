// Library-internal entry points shared by api.hip, comm.hip and multi.hip.  NOT part of the C-ABI (include/comorag_hip.h).
#pragma once
#include <stdint.h>

#include "../../include/comorag_hip.h"

// sets the calling thread's error message (cmr_last_error) and returns `code`
int cmr_fail(int code, const char* fmt, ...);

// cmr_index_search in two halves: `begin` enqueues the whole search on a workspace stream of the index and returns at once,
// `finish` waits for it, reports a non-finite query and copies ids / scores / min / max out (any of them may be NULL).  Every
// pending search must be finished or abandoned.  take_lock = true holds the index's shared lock from begin to finish — both
// calls must then come from ONE thread; take_lock = false is for an owner that excludes appends itself (the multi-device index
// holds its own layout lock around begin .. finish and is the only one to append to its shards).
struct CmrPending;
int cmr_index_search_begin(cmr_index_t* idx, const float* q, int nq, int k, const float* min_score, bool take_lock, CmrPending** out);
int cmr_index_search_finish(CmrPending* p, int64_t* out_ids, float* out_scores, float* out_min, float* out_max);
void cmr_index_search_abandon(CmrPending* p);

// shrink an index to its first n_rows rows (roll-back of a multi-shard append that failed on a later shard)
int cmr_index_truncate(cmr_index_t* idx, long long n_rows);
