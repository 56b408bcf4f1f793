// host_pool.cpp -- see host_pool.hpp.
#include "host_pool.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace hmrm {
namespace {

// One job at a time: `pieces` contiguous ranges handed out through one atomic ticket that carries the job's
// generation in its high half, so a helper that wakes late (its job already finished, the next one being posted)
// can never take a piece of a job whose shape it did not read: a ticket is claimed by compare-and-swap against
// the generation the helper copied, under the lock, together with fn / n / pieces.  The caller takes pieces too and
// then waits for the helpers that are still inside one.  Helpers sleep on a condition variable between jobs (after
// a short spin: frames of a sequence arrive ~0.1 ms apart).
struct Pool {
	std::mutex mu;               // guards generation / fn / shape of the current job
	std::condition_variable cv;
	std::mutex job_mu;           // serialises callers
	uint64_t generation = 0;
	const std::function<void(int, int)> *fn = nullptr;
	int n = 0, pieces = 0;
	std::atomic<uint64_t> ticket{0}; // (generation << 32) | next piece
	std::atomic<int> done{0};
	std::atomic<uint64_t> posted{0}; // (lock-free copy of generation for the spin)
	int helpers = 0;

	void run_pieces(uint64_t gen, const std::function<void(int, int)> *f, int count, int parts) {
		for (;;) {
			uint64_t t = ticket.load(std::memory_order_acquire);
			int p;
			for (;;) {
				if ((t >> 32) != (gen & 0xffffffffull)) return; // another job's counter: not ours to touch
				p = (int)(t & 0xffffffffull);
				if (p >= parts) return;
				if (ticket.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) break;
			}
			const int b = (int)((int64_t)count * p / parts), e = (int)((int64_t)count * (p + 1) / parts);
			(*f)(b, e);
			done.fetch_add(1, std::memory_order_acq_rel);
		}
	}

	void helper_main() {
		uint64_t seen = 0;
		for (;;) {
			// brief spin, then sleep
			for (int i = 0; i < 2000 && posted.load(std::memory_order_acquire) == seen; ++i) {
#if defined(__x86_64__)
				__builtin_ia32_pause();
#endif
			}
			const std::function<void(int, int)> *f;
			int count, parts;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return generation != seen; });
				seen = generation;
				f = fn;
				count = n;
				parts = pieces;
			}
			run_pieces(seen, f, count, parts);
		}
	}
};

Pool *pool() {
	// (never destroyed: helpers may be asleep inside it when the process exits)
	static Pool *p = [] {
		Pool *q = new Pool();
		int want = 3;
		if (const char *s = getenv("HMRM_HOST_THREADS")) want = atoi(s) - 1;
		const unsigned hw = std::thread::hardware_concurrency();
		if (hw > 0 && (int)hw - 1 < want) want = (int)hw - 1;
		if (want < 0) want = 0;
		if (want > 15) want = 15;
		for (int i = 0; i < want; ++i) {
			try {
				std::thread(&Pool::helper_main, q).detach();
				++q->helpers;
			} catch (...) {
				break;
			}
		}
		return q;
	}();
	return p;
}

} // namespace

int host_pool_helpers() { return pool()->helpers; }

void parallel_ranges(int n, int grain, const std::function<void(int, int)> &fn) {
	if (n <= 0) return;
	Pool *p = pool();
	if (grain < 1) grain = 1;
	int pieces = n / grain;
	if (pieces > p->helpers + 1) pieces = p->helpers + 1;
	if (pieces <= 1) {
		fn(0, n);
		return;
	}
	std::lock_guard<std::mutex> job(p->job_mu);
	{
		std::lock_guard<std::mutex> lk(p->mu);
		p->fn = &fn;
		p->n = n;
		p->pieces = pieces;
		p->done.store(0, std::memory_order_release);
		++p->generation;
		p->ticket.store((p->generation & 0xffffffffull) << 32, std::memory_order_release);
		p->posted.store(p->generation, std::memory_order_release);
	}
	p->cv.notify_all();
	p->run_pieces(p->generation, &fn, n, pieces);
	while (p->done.load(std::memory_order_acquire) < pieces) {
#if defined(__x86_64__)
		__builtin_ia32_pause();
#endif
	}
}

} // namespace hmrm
