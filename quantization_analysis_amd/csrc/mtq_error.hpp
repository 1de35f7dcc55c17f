// mtq_error.hpp — thread-local error reporting shared by the translation units of libmtq_hip.so.
#pragma once
#include "../../include/mtq.h"

namespace mtq {
// Records `msg` as the calling thread's last error and returns `code`.
int fail(int code, const char *msg);
int failf(int code, const char *fmt, ...);
// MTQ_OK when a HIP device is usable by this process, MTQ_ERR_NO_DEVICE otherwise (cached).
int require_device();
// Zeroed device counters for one K1 launch (persistent waves claim their units from them): kWorkGroups counters,
// kWorkStride unsigneds apart.  Slots come from a per-device ring of kWorkSlots; the follow-up kernel of the launch
// (tile_stats_redo_flagged) sets the slot back to zero, so no memset sits on the stream.  nullptr when the ring cannot
// be allocated.
constexpr int kWorkGroups = 64, kWorkStride = 32, kWorkSlots = 128;
// Word 1 of a slot carries the id of the last launch that marked a tile for the literal fix-up (ids are unique per launch,
// so the word never needs resetting): the follow-up kernel returns at once unless it finds its own launch's id there.
constexpr int kWorkStamp = 1;
unsigned next_launch_id();
unsigned *work_counter_slot();
// hipGetLastError() → MTQ_OK / MTQ_ERR_HIP with the kernel name in the message.
int check_launch(const char *what);
} // namespace mtq
