// sequencer.hip — step programs (see sequencer.hpp) and the recordable stream operations of the C ABI.
#include "sequencer.hpp"

namespace fnr {
namespace seq {

thread_local fnr_program* g_recording = nullptr;

void push(const char* name, OpFn fn) {
  if (g_recording) g_recording->ops.push_back(Op{name, std::move(fn)});
}

void poison(const char* name) {
  if (g_recording && !g_recording->poisoned) g_recording->poisoned = name;
}

// fnr_stream_wait_stream: a small ring of events per thread.  An event may be re-recorded while an earlier wait on it is
// still pending — hipStreamWaitEvent captures the record that precedes it.
static const int WAIT_RING = 16;
static thread_local hipEvent_t g_ring[WAIT_RING];
static thread_local int g_ring_n = 0, g_ring_next = 0;

static int ring_event(hipEvent_t* out) {
  if (g_ring_n < WAIT_RING) {
    FNR_HIP(hipEventCreateWithFlags(&g_ring[g_ring_n], hipEventDisableTiming));
    ++g_ring_n;
  }
  *out = g_ring[g_ring_next];
  g_ring_next = (g_ring_next + 1) % g_ring_n;
  return FNR_OK;
}

}  // namespace seq
}  // namespace fnr

using namespace fnr;

extern "C" int fnr_program_create(fnr_program** out) {
  FNR_CHECK_ARG(out, "program_create: null argument");
  *out = new fnr_program();
  return FNR_OK;
}

extern "C" int fnr_program_destroy(fnr_program* p) {
  if (p && seq::g_recording == p) seq::g_recording = nullptr;
  delete p;
  return FNR_OK;
}

extern "C" int fnr_program_begin(fnr_program* p) {
  FNR_CHECK_ARG(p, "program_begin: null program");
  FNR_CHECK_ARG(seq::g_recording == nullptr, "program_begin: this thread is already recording a program");
  p->ops.clear();
  p->poisoned = nullptr;
  p->recording = true;
  seq::g_recording = p;
  return FNR_OK;
}

extern "C" int fnr_program_end(fnr_program* p) {
  FNR_CHECK_ARG(p && seq::g_recording == p, "program_end: this thread is not recording that program");
  seq::g_recording = nullptr;
  p->recording = false;
  if (p->poisoned) {
    p->ops.clear();
    set_error("step program: %s ran while recording and cannot be replayed", p->poisoned);
    return FNR_ERR_UNSUPPORTED;
  }
  return FNR_OK;
}

extern "C" int fnr_program_abort(fnr_program* p) {
  if (p && seq::g_recording == p) seq::g_recording = nullptr;
  if (p) {
    p->recording = false;
    p->ops.clear();
  }
  return FNR_OK;
}

extern "C" int64_t fnr_program_size(const fnr_program* p) { return p ? (int64_t)p->ops.size() : -1; }

extern "C" const char* fnr_program_op_name(const fnr_program* p, int64_t i) {
  return (p && i >= 0 && i < (int64_t)p->ops.size()) ? p->ops[(size_t)i].name : nullptr;
}

extern "C" int fnr_program_replay(const fnr_program* p, const fnr_step_scalars* scalars) {
  FNR_CHECK_ARG(p && !p->recording, "program_replay: null program, or one that is being recorded");
  FNR_CHECK_ARG(seq::g_recording == nullptr, "program_replay: this thread is recording a program");
  for (const seq::Op& op : p->ops) {
    const int rc = op.run(scalars);
    if (rc != FNR_OK) return rc;   // (the entry point has set the message)
  }
  return FNR_OK;
}

extern "C" int fnr_event_create(void** event_out) {
  FNR_CHECK_ARG(event_out, "event_create: null argument");
  hipEvent_t e;
  FNR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event_out = e;
  return FNR_OK;
}

extern "C" int fnr_event_destroy(void* event) {
  if (event) FNR_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
  return FNR_OK;
}

extern "C" int fnr_event_record(void* event, void* stream) {
  FNR_CHECK_ARG(event, "event_record: null event");
  if (seq::recording()) seq::push("fnr_event_record", [=](const fnr_step_scalars*) { return fnr_event_record(event, stream); });
  FNR_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
  return FNR_OK;
}

extern "C" int fnr_stream_wait_event(void* stream, void* event) {
  FNR_CHECK_ARG(event, "stream_wait_event: null event");
  if (seq::recording())
    seq::push("fnr_stream_wait_event", [=](const fnr_step_scalars*) { return fnr_stream_wait_event(stream, event); });
  FNR_HIP(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
  return FNR_OK;
}

extern "C" int fnr_stream_wait_stream(void* waiting, void* signalling) {
  if (seq::recording())
    seq::push("fnr_stream_wait_stream", [=](const fnr_step_scalars*) { return fnr_stream_wait_stream(waiting, signalling); });
  if (waiting == signalling) return FNR_OK;
  hipEvent_t e;
  const int rc = seq::ring_event(&e);
  if (rc != FNR_OK) return rc;
  FNR_HIP(hipEventRecord(e, as_stream(signalling)));
  FNR_HIP(hipStreamWaitEvent(as_stream(waiting), e, 0));
  return FNR_OK;
}
