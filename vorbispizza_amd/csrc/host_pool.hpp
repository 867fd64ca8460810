// A small fork-join pool for the host half of vpz_decoder_synth: the per-packet state machine of a large batch is
// split over the host cores (vpz_decoder.hip, run_state_machine_parallel).  Workers spin for a few microseconds
// before they block, so that the two fork-joins of one call do not each pay a futex wake-up.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vpz {

class HostPool {
public:
    // spin_us: how long an idle worker polls before it blocks (VPZ_HOST_SPIN_US; a host that calls the decoder back to back
    // -- a call every few hundred microseconds -- saves a futex wake-up per fork by polling through the gap, at the price of
    // busy cores; the default only bridges the forks of ONE call)
    explicit HostPool(int parties, int spin_us = 50) : parties_(parties < 1 ? 1 : parties), spin_us_(spin_us < 0 ? 0 : spin_us)
    {
        for (int i = 1; i < parties_; ++i) workers_.emplace_back([this, i] { worker(i); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    HostPool(const HostPool &) = delete;
    HostPool &operator=(const HostPool &) = delete;

    int parties() const { return parties_; }

    // fn(i) for every i in [0, parties()); the caller runs i = 0.  Returns when all are done: true, or false when any
    // party's share threw (std::bad_alloc from a vector, ...) -- an exception must neither leave a worker thread
    // (std::terminate) nor unwind through the extern "C" entry the caller sits in; the caller reports VPZ_E_NOMEM.
    bool run(const std::function<void(int)> &fn)
    {
        failed_.store(false, std::memory_order_relaxed);
        if (parties_ == 1) { guarded(fn, 0); return !failed_.load(std::memory_order_relaxed); }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn;
            pending_.store(parties_ - 1, std::memory_order_relaxed);
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        guarded(fn, 0);
        // the workers' share is as long as the caller's: spin, it ends within microseconds
        while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        fn_ = nullptr;
        return !failed_.load(std::memory_order_acquire);
    }

private:
    void guarded(const std::function<void(int)> &fn, int index)
    {
        try {
            fn(index);
        } catch (...) {
            failed_.store(true, std::memory_order_release);
        }
    }

    void worker(int index)
    {
        unsigned seen = 0;
        for (;;) {
            // short spin first: the second fork of a call follows the first within tens of microseconds
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us_)) {
                if (generation_.load(std::memory_order_acquire) != seen) { got = true; break; }
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen; });
            }
            seen = generation_.load(std::memory_order_acquire);
            if (stop_) return;
            const std::function<void(int)> *fn;
            {
                std::lock_guard<std::mutex> lk(m_);  // pairs with run(): fn_ is published under the lock
                fn = fn_;
            }
            if (fn) guarded(*fn, index);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }

    const int parties_;
    const int spin_us_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<unsigned> generation_{0};
    std::atomic<int> pending_{0};
    std::atomic<bool> failed_{false};
    const std::function<void(int)> *fn_ = nullptr;
    bool stop_ = false;
};

}  // namespace vpz
