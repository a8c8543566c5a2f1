// scalce_hip.hip -- C ABI (include/scalce_hip.h) over the gfx950 kernels.
// Host side of the hot path: owns the device buffers of a shard, sequences the kernels on the
// caller's stream and reads back the handful of counters that size the next launches.
// There is NO CPU fallback: every stage runs on the device or returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"
#include "automaton.hpp"
#include "kernels_acl.hpp"
#include "kernels_fastq.hpp"

using namespace scalce;

// The host side by stage (one translation unit: the kernels are templates and inline functions of the kernels_*.hpp headers,
// and everything below shares the state declared in host_state.inc):
#include "host_state.inc"
#include "host_ingest.inc"
#include "host_tokenize.inc"
#include "host_order_emit.inc"
#include "host_entropy.inc"
#include "host_api.inc"
#include "host_decode.inc"
