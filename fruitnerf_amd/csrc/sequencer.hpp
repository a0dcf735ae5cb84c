// sequencer.hpp — step programs: the launch sequence of a training step, recorded once at the C ABI and replayed natively.
//
// A training step of the default loop (fruitnerf_amd/training.py::TrainingSteps) is ~30 entry-point calls on two HIP
// streams with four cross-stream dependencies; every pointer in it is stable from step to step (parameter arenas,
// persistent workspaces, the step arena of per-step buffers), and only a handful of scalars change: the learning rates
// and step counts of the optimiser groups, the sampler's anneal, the random-number counter, where the five loss values
// go.  While a program is being recorded (fnr_program_begin .. fnr_program_end on the calling thread) every recordable
// entry point appends a closure over its own arguments — host structs and host arrays copied by value — and then runs as
// usual; fnr_program_replay() calls the closures in order with the per-step scalars patched in.  Same entry points, same
// arguments, same streams, same order: a replayed step is the recorded step (tests/test_gpu_sequencer.py), minus the
// interpreter (0.53 -> ~0.2 ms of host time per step: what is left is the HIP runtime's own launch cost).
// No counterpart in the reference (its loop is nerfstudio's Python Trainer).
#pragma once
#include <array>
#include <functional>
#include <vector>

#include "common.hpp"

namespace fnr {
namespace seq {

typedef std::function<int(const fnr_step_scalars*)> OpFn;

struct Op {
  const char* name;   // the entry point's name (static string)
  OpFn run;
};

}  // namespace seq
}  // namespace fnr

struct fnr_program {
  std::vector<fnr::seq::Op> ops;
  bool recording = false;
  const char* poisoned = nullptr;   // the entry point that ran while recording and cannot be replayed (FNR_SEQ_UNRECORDABLE)
};

namespace fnr {
namespace seq {

extern thread_local fnr_program* g_recording;

static inline bool recording() { return g_recording != nullptr; }
void push(const char* name, OpFn fn);
void poison(const char* name);

// lr / step of a recorded optimiser struct come from the replay's scalars when the struct names a slot
static inline fnr_table_adam patched(fnr_table_adam a, const fnr_step_scalars* s) {
  if (s && a.slot >= 1 && a.slot <= FNR_PROGRAM_ADAM_SLOTS && s->adam[a.slot - 1].step > 0) {
    a.lr = s->adam[a.slot - 1].lr;
    a.step = s->adam[a.slot - 1].step;
  }
  return a;
}

template <class T, int N>
static inline std::array<T, N> copy_n(const T* src, int n) {
  std::array<T, N> out{};
  for (int i = 0; i < n && i < N; ++i) out[i] = src ? src[i] : T{};
  return out;
}

}  // namespace seq
}  // namespace fnr

// An entry point that enqueues device work and has no recording hook: a program recorded across it would silently skip it.
#define FNR_SEQ_UNRECORDABLE(name)                          \
  do {                                                      \
    if (::fnr::seq::recording()) ::fnr::seq::poison(name);  \
  } while (0)
