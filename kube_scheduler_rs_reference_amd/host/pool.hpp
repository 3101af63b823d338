// pool.hpp -- the host mirror's worker threads and its phase clock.
//
// The per-pod string work around the kernel (quantity parsing, dictionary lookups, binding objects) is fanned out over threads; a C3-size
// batch (100 000 pods) passes through three such regions, and starting 32 threads three times per batch cost more than the work of some
// regions (round 6: `std::thread` construction is ~40 us each on the 256-thread GPU box, paid serially by the spawning thread).  One
// process-wide pool, started on first use, parked on a condition variable between regions.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <functional>
#include <memory>
#include <new>
#include <type_traits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

namespace ksched_host {

// worker threads for the per-pod string work (KSCHED_HOST_THREADS overrides the hardware's count: 1 = everything on the caller's thread)
inline uint32_t host_threads() {
    if (const char *e = std::getenv("KSCHED_HOST_THREADS")) return std::max(1u, (uint32_t)std::strtoul(e, nullptr, 0));
    return std::max(1u, std::thread::hardware_concurrency());
}

class WorkerPool {
public:
    static WorkerPool &instance() {
        static WorkerPool p;
        return p;
    }
    // fn(lo, hi, part) over [0, n) cut into `parts` contiguous ranges (part = index of the range); the caller's thread takes part 0 and
    // returns when every part is done.  An exception in a part is rethrown here (the lowest part's), after all parts have finished.
    // Regions do not nest: a region started from inside a part runs on the calling thread.
    void run(size_t n, uint32_t parts, const std::function<void(size_t, size_t, uint32_t)> &fn) {
        parts = (uint32_t)std::min<size_t>(std::max<size_t>(parts, 1), std::max<size_t>(n, 1));
        if (parts <= 1 || in_region_) {
            fn(0, n, 0);
            return;
        }
        std::unique_lock<std::mutex> region(region_mu_);  // one region at a time (several host threads may share the pool)
        ensure(parts - 1);
        errors_.assign(parts, nullptr);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            parts_ = parts;
            next_part_ = 1;
            pending_ = parts - 1;
            ++generation_;
        }
        cv_.notify_all();
        in_region_ = true;
        try {
            fn(0, n / parts, 0);
        } catch (...) {
            errors_[0] = std::current_exception();
        }
        in_region_ = false;
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_.wait(lk, [&] { return pending_ == 0; });
            fn_ = nullptr;
        }
        for (auto &e : errors_)
            if (e) std::rethrow_exception(e);
    }
    // how many parts a region over `n` items should have: at least `grain` items per part, at most `cap` parts and the host's threads
    static uint32_t parts_for(size_t n, size_t grain = 1024, uint32_t cap = 32) {
        return (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)host_threads(), (size_t)cap, n / std::max<size_t>(grain, 1)}));
    }

    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            ++generation_;
        }
        cv_.notify_all();
        if (pid_ == ::getpid())
            for (auto &t : threads_) t.join();
        else
            new (&threads_) std::vector<std::thread>();  // (a forked child: nothing to join)
    }

private:
    WorkerPool() = default;
    void ensure(uint32_t workers) {
        if (pid_ != ::getpid()) {  // a forked child inherits the object but none of the threads: start over (the old handles are abandoned, never joined)
            new (&threads_) std::vector<std::thread>();
            pid_ = ::getpid();
        }
        while (threads_.size() < workers) threads_.emplace_back([this] { loop(); });
    }
    void loop() {
        uint64_t seen = 0;
        in_region_ = true;  // (a worker never starts a nested region)
        for (;;) {
            const std::function<void(size_t, size_t, uint32_t)> *fn = nullptr;
            size_t n = 0;
            uint32_t parts = 0, part = 0;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || (generation_ != seen && fn_ && next_part_ < parts_); });
                if (stop_) return;
                fn = fn_;
                n = n_;
                parts = parts_;
                part = next_part_++;
                if (next_part_ >= parts_) seen = generation_;  // nothing left in this region for anybody
            }
            try {
                (*fn)(n * part / parts, n * (part + 1) / parts, part);
            } catch (...) {
                errors_[part] = std::current_exception();
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex region_mu_, mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::vector<std::exception_ptr> errors_;
    const std::function<void(size_t, size_t, uint32_t)> *fn_ = nullptr;
    size_t n_ = 0;
    uint32_t parts_ = 0, next_part_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
    pid_t pid_ = ::getpid();
    static thread_local bool in_region_;
};
inline thread_local bool WorkerPool::in_region_ = false;

// Big temporaries of a batch (100 000 outcomes with their strings, the staged snapshot update's keys, the draws) take milliseconds to
// free; the caller of reconcile_batch should not wait for that.  discard_later() hands an object to a process-wide reaper thread, which
// destroys it; at most a few batches' worth are ever queued (a full queue destroys on the spot).
class Reaper {
public:
    static Reaper &instance() {
        static Reaper r;
        return r;
    }
    template <class T>
    void discard_later(T &&object) {
        auto box = std::make_shared<std::decay_t<T>>(std::forward<T>(object));
        std::function<void()> drop = [box]() mutable { box.reset(); };
        box.reset();
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (queue_.size() < 64) {
                queue_.push_back(std::move(drop));
                drop = nullptr;
            }
        }
        if (drop) drop();  // (queue full: on the caller's thread after all)
        else cv_.notify_one();
    }
    ~Reaper() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (thread_.joinable()) thread_.join();
    }

private:
    Reaper() : thread_([this] { loop(); }) {}
    void loop() {
        for (;;) {
            std::vector<std::function<void()>> batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty() && stop_) return;
                batch.swap(queue_);
            }
            for (auto &f : batch) f = nullptr;  // the captured objects die here
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::function<void()>> queue_;
    bool stop_ = false;
    std::thread thread_;
};

// Where a batch's host time goes: KSCHED_HOST_TIMING=2 prints one line per phase to stderr (tools/host_loop.py --phases).
class PhaseClock {
public:
    explicit PhaseClock(const char *what) : what_(what), on_(level() >= 2), t_(std::chrono::steady_clock::now()) {}
    void lap(const char *phase) {
        if (!on_) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "  phase %s / %s: %.3f ms\n", what_, phase, std::chrono::duration<double, std::milli>(now - t_).count());
        t_ = now;
    }
    static int level() {
        static const int l = [] {
            const char *e = std::getenv("KSCHED_HOST_TIMING");
            return e ? std::max(1, std::atoi(e)) : 0;
        }();
        return l;
    }

private:
    const char *what_;
    bool on_;
    std::chrono::steady_clock::time_point t_;
};

}  // namespace ksched_host
