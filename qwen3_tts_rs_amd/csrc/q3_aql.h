// q3_aql.h — the frame loop's own AQL submission path (internal to libq3tts.so; see q3_aql.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace q3 {

struct AqlProgram;

// Per-node policy of a program, chosen by the engine when it converts a captured frame:
//   fence: 1 = agent-scope acquire / release at the kernel boundary (what HIP writes), 0 = none — legal only between nodes
//          whose cross-kernel traffic is write-through (sc1) on the producer and L1-bypassing (sc1) on the consumer, which the
//          product kernels are NOT (plain loads / stores): 0 is a probe setting that yields wrong results on them.
struct AqlPolicy { int fence = 1; int acquire = 1, release = 1; };      // acquire / release: the two halves of `fence`, set separately by probes

// Converts a captured, purely linear kernel graph into a packet program for `device`. Returns nullptr and fills *why when
// the graph holds anything the converter does not handle (the caller then stays on hipGraphLaunch).
AqlProgram* aql_program_create(hipGraph_t graph, int device, const AqlPolicy& pol, std::string* why);
void aql_program_destroy(AqlProgram* p);
int aql_program_nodes(const AqlProgram* p);

// Enqueues `frames` replays of the program behind everything this program submitted before. The caller has synchronised the
// HIP stream whose results the first node reads. Returns false with *why on failure; *submitted = whole frames handed to the
// device (== frames on success; on a ring time-out the frames already written, after which the program is dead: every later
// submit / wait fails at once and destroy does not block).
bool aql_submit(AqlProgram* p, int frames, std::string* why, int* submitted = nullptr);
// Blocks until every submitted replay has completed (system-scope release: the results are visible to HIP streams and host).
bool aql_wait(AqlProgram* p, std::string* why);

}  // namespace q3
