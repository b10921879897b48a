// worker_pool.h -- the fork-join pool the CLIs' host side uses (ingest.h, par_inflate.h)
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hast {

// ------------------------------------------------------------------------------------------------
// minimal fork-join pool: run(fn) executes fn(worker_index) on every worker and waits
// ------------------------------------------------------------------------------------------------
class WorkerPool {
  public:
    explicit WorkerPool(int n) : n_(n < 1 ? 1 : n) {
        for (int i = 1; i < n_; ++i) threads_.emplace_back([this, i] { loop(i); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    int size() const { return n_; }
    void run(const std::function<void(int)> &fn) {
        if (n_ == 1) { fn(0); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(mu_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

  private:
    void loop(int idx) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *fn;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            (*fn)(idx);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

}  // namespace hast
