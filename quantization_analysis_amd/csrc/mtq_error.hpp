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
// (tile_stats_redo_flagged) sets the slot back to zero, so no memset sits on the stream.  A slot is handed out again only
// behind the event recorded after that follow-up kernel (mtq_slot_ring.hpp): more than kWorkSlots pending launches queue
// instead of sharing counters.
constexpr int kWorkGroups = 64, kWorkStride = 32, kWorkSlots = 128;
// Word 1 of a slot carries the id of the last launch that marked a tile for the literal fix-up (ids are unique per launch,
// so the word never needs resetting): the follow-up kernel returns at once unless it finds its own launch's id there.
constexpr int kWorkStamp = 1;
// Word 2 of a slot counts the waves of the launch that have finished (exact-integer bf16 kernel): the wave that completes the count sets
// the unit counters and this word back to zero, so that launch needs no follow-up kernel for the reset (round 3).
constexpr int kWorkDone = 2;
unsigned next_launch_id();
struct WorkSlot { unsigned *counters = nullptr; int index = -1, device = -1; };
// MTQ_OK and a slot whose previous user `stream` now waits for, or MTQ_ERR_HIP.
int work_counter_acquire(void *stream, WorkSlot *out);
// Records the slot's event on `stream` (behind the follow-up kernel) and gives the slot back.
void work_counter_release(const WorkSlot &slot, void *stream);
// The launch the slot was acquired for could not be enqueued: the slot goes back without a new event.
void work_counter_abandon(const WorkSlot &slot);
// hipGetLastError() → MTQ_OK / MTQ_ERR_HIP with the kernel name in the message.
int check_launch(const char *what);
// mtq_shutdown's parts (each translation unit releases what it owns; none of them runs from a static destructor)
void scan_shutdown();       // mtq_scan.hip: the device jump-ahead tables
void host_shutdown();       // mtq_host.cpp: the scan thread pool
} // namespace mtq
