// q3_aql.h — the frame loop's own AQL submission path (internal to libq3tts.so; see q3_aql.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace q3 {

struct AqlProgram;

// Fence policy of a program, chosen by the engine when it converts a captured frame.
//   acquire / release: 1 = agent-scope fence at the kernel boundary (what HIP writes on every packet), 0 = none. A node may go
//   without its RELEASE fence only if everything it stores for later nodes is write-through and drained, and without its ACQUIRE
//   fence only if everything it loads from earlier nodes of the same frame bypasses the vector L1 and it fetches none of it
//   through the scalar cache (q3_kernels.h "activation transport").
//   node_policy (optional): called per kernel node with the kernel's (mangled) name; it may clear *acquire / *release for
//   kernels that keep to that rule. The first node of a frame always acquires and the last always releases at agent scope,
//   whatever the policy says: state that crosses frames (positions, counters, the talker's K/V) uses plain accesses.
struct AqlPolicy {
    int fence = 1; int acquire = 1, release = 1;      // program-wide defaults; 0 here is a probe setting (wrong results with plain-access kernels)
    void (*node_policy)(const char* kernel_name, int* acquire, int* release) = nullptr;
};

// Converts a captured, purely linear kernel graph into a packet program for `device`. Returns nullptr and fills *why when
// the graph holds anything the converter does not handle (the caller then stays on hipGraphLaunch).
AqlProgram* aql_program_create(hipGraph_t graph, int device, const AqlPolicy& pol, std::string* why);
void aql_program_destroy(AqlProgram* p);
int aql_program_nodes(const AqlProgram* p);
void aql_program_fence_free(const AqlProgram* p, int* acquire_free, int* release_free);    // nodes without their acquire / release fence

// Enqueues `frames` replays of the program behind everything this program submitted before. The caller has synchronised the
// HIP stream whose results the first node reads. Returns false with *why on failure; *submitted = whole frames handed to the
// device (== frames on success; on a ring time-out the frames already written, after which the program is dead: every later
// submit / wait fails at once and destroy does not block).
bool aql_submit(AqlProgram* p, int frames, std::string* why, int* submitted = nullptr);
// Blocks until every submitted replay has completed (system-scope release: the results are visible to HIP streams and host).
bool aql_wait(AqlProgram* p, std::string* why);
// Confines the queue the program submits to (one per device: every program on it) to the first `first_cus` bits of the CU mask —
// first_cus / 8 CUs of each XCD; first_cus <= 0 gives the whole chip back. Calls nest (a count); the caller has waited for its replays.
bool aql_restrict_cus(AqlProgram* p, int first_cus, std::string* why);

}  // namespace q3
