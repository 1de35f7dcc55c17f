// mtq_slot_ring.hpp — bookkeeping of the per-device ring of K1 work-counter slots (mtq_error.hpp), host side only.
//
// A slot is handed to one K1 launch; that launch's follow-up kernel sets the slot's counters back to zero on the same
// stream.  A slot may therefore only be handed out again to a launch that is ordered AFTER that follow-up kernel:
//   acquire(stream)  picks the next slot that no thread holds between its acquire and its release, and makes `stream`
//                    wait (device side, hipStreamWaitEvent) for the event its previous user recorded;
//   release(i, stream) records the slot's event behind the follow-up kernel and gives the slot back.
// With more than N launches pending the (N+1)-th simply queues behind the first one's follow-up kernel instead of
// sharing its counters.  The event operations are a template parameter so that the bookkeeping is unit-tested on a host
// without a GPU (mtq_selftest_slot_ring, tests/test_capi_host.py).
#pragma once
#include <atomic>
#include <mutex>

namespace mtq {

template <typename Ops, int N>
class SlotRing {
public:
    // → slot index, or −1 (every slot is held by a thread between acquire and release, or no slot's event could be waited for)
    int acquire(typename Ops::stream s)
    {
        for (int tries = 0; tries < N; ++tries) {
            const int i = (int)(next_.fetch_add(1u, std::memory_order_relaxed) % (unsigned)N);
            Slot &sl = slots_[i];
            std::lock_guard<std::mutex> lock(sl.mu);
            if (sl.held) continue;                       // another thread is between acquire and release on it
            if (sl.recorded && !Ops::wait(s, sl.ev)) continue;   // no ordering to be had behind this slot's previous user: try the next one
            sl.held = true;
            return i;
        }
        return -1;
    }
    // Call after the follow-up kernel has been enqueued on `s` (also when that enqueue failed and a memset took its place).
    bool release(int i, typename Ops::stream s)
    {
        Slot &sl = slots_[i];
        std::lock_guard<std::mutex> lock(sl.mu);
        bool ok = true;
        if (!sl.created) { ok = Ops::create(sl.ev); sl.created = ok; }
        if (ok) ok = Ops::record(sl.ev, s);
        sl.recorded = ok;                                // a slot whose event could not be recorded has no ordering to offer
        sl.held = false;
        return ok;
    }

    // The launch the slot was acquired for never happened (its enqueue failed): give the slot back as it was — whatever its
    // previous user recorded still orders its next user.
    void abandon(int i)
    {
        Slot &sl = slots_[i];
        std::lock_guard<std::mutex> lock(sl.mu);
        sl.held = false;
    }
    // mtq_shutdown: every slot back to its fresh state, its event destroyed.  The caller has drained the device.
    void reset()
    {
        for (int i = 0; i < N; ++i) {
            Slot &sl = slots_[i];
            std::lock_guard<std::mutex> lock(sl.mu);
            if (sl.created) Ops::destroy(sl.ev);
            sl.created = sl.recorded = sl.held = false;
        }
    }

private:
    struct Slot {
        std::mutex mu;
        typename Ops::event ev{};
        bool created = false, recorded = false, held = false;
    };
    Slot slots_[N];
    std::atomic<unsigned> next_{0u};
};

} // namespace mtq
