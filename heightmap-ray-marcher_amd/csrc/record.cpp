// record.cpp -- recording driver: a programmatic camera sweep rendered frame by
// frame and saved as PNGs, the headless counterpart of the reference's recording
// mode (main/hmap.cpp:642-649, 869-900, 1131-1144).  The reference leaves the
// per-frame parameter change to the user ("For a programmatic animation, alter
// this block and recompile", hmap.cpp:907-926, an empty stub); the sweep built
// here is the orbit of BASELINE config C5.  File naming follows
// hmap.cpp:1132-1134: <dir>/hmap_<id>_<n>.png, n = 0 .. frames-1.
//
// Frames are independent, so the sweep shards over devices the way BASELINE config C5
// says: frame k is rendered by scene k mod N (one scene per GPU, maps replicated), no
// exchange between devices.  Per scene one host thread keeps a few frames in flight
// through the asynchronous ring of api.cpp (kernel k+1 overlaps the PCIe copy of frame
// k); finished frames stay in the ring's pinned memory and are lent to a shared pool of
// PNG encoder threads (the stb-identical encoder needs ~0.3 s per 4K frame on one
// core, the GPU 0.25 ms), which give the slot back when the file is written.  Memory is
// bounded by the ring: at most `slots` raw frames per scene exist, whatever the number
// of encoder threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hmrm.h"
#include "image_io.hpp"

namespace hmrm {
int set_error(int code, const char *msg);
}

extern "C" {

// Camera of frame `frame` of `frames` on a horizontal circle of `radius` around
// (centre_x, centre_y), always looking at the centre: hang_k = hang0 + 2*pi*k/frames,
// pos_k = centre - radius * (cos hang_k, sin hang_k); height, vang and everything else
// are taken from `base`.
void hmrm_orbit_camera(const hmrm_camera *base, double centre_x, double centre_y, double radius,
                       double hang0, int32_t frame, int32_t frames, hmrm_camera *out) {
	*out = *base;
	const double turn = (2.0 * M_PI * (double)frame) / (double)(frames > 0 ? frames : 1);
	const double hang = hang0 + turn;
	out->hang = hang;
	out->pos[0] = centre_x - radius * std::cos(hang);
	out->pos[1] = centre_y - radius * std::sin(hang);
}

// Which of n_devices scenes renders orbit frame `frame` (BASELINE config C5: frame k -> GPU k mod N).
int32_t hmrm_orbit_frame_owner(int32_t frame, int32_t n_devices) {
	return n_devices > 0 && frame >= 0 ? frame % n_devices : -1;
}

int hmrm_record_orbit_multi(hmrm_scene *const *scenes, int32_t n_scenes, const hmrm_camera *base, double centre_x,
                            double centre_y, double radius, double hang0, int32_t frames, const char *dir,
                            long long id, int32_t encoder_threads, int32_t verbose) {
	if (!scenes || n_scenes <= 0 || !base || !dir || frames <= 0)
		return hmrm::set_error(HMRM_E_ARG, "hmrm_record_orbit: bad argument");
	for (int i = 0; i < n_scenes; ++i)
		if (!scenes[i]) return hmrm::set_error(HMRM_E_ARG, "hmrm_record_orbit: NULL scene");
	if (base->width <= 0 || base->height <= 0) return hmrm::set_error(HMRM_E_ARG, "resolution must be positive");
	const size_t W = (size_t)base->width, H = (size_t)base->height;
	const size_t frame_bytes = W * H * 4;
	const int n = std::min(n_scenes, frames);

	// encoder threads: one per host core by default, at most 64 and at most one per frame
	unsigned nthreads = encoder_threads > 0 ? (unsigned)encoder_threads : std::thread::hardware_concurrency();
	if (nthreads == 0) nthreads = 4;
	nthreads = std::min(nthreads, encoder_threads > 0 ? 1024u : 64u);
	nthreads = std::min(nthreads, (unsigned)frames);
	// ring slots per scene: enough raw frames in flight to keep every encoder busy plus two being
	// rendered / copied, within a pinned-memory budget of 4 GiB over all scenes (never less than 2)
	size_t slots = (nthreads + (size_t)n - 1) / (size_t)n + 2;
	const size_t budget_frames = ((size_t)4 << 30) / (frame_bytes ? frame_bytes : 1) / (size_t)n;
	slots = std::max<size_t>(2, std::min(std::min<size_t>(slots, 60), budget_frames));

	struct Job {
		int index;
		int scene;
		int32_t ticket;
		const uint8_t *pixels;
	};
	std::mutex mu;
	std::condition_variable cv_job, cv_slot;
	std::deque<Job> queue;
	std::vector<size_t> free_slots((size_t)n, slots);
	int producers_left = n;
	std::atomic<int> failures(0);
	std::atomic<int> first_error(HMRM_OK);
	std::string first_error_text;
	const std::string prefix = std::string(dir) + "/hmap_" + std::to_string(id) + "_";

	auto note_error = [&](int rc) {
		int expected = HMRM_OK;
		if (first_error.compare_exchange_strong(expected, rc)) {
			std::lock_guard<std::mutex> lk(mu);
			first_error_text = hmrm_last_error();
		}
	};

	auto encoder = [&]() {
		for (;;) {
			Job job;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_job.wait(lk, [&] { return producers_left == 0 || !queue.empty(); });
				if (queue.empty()) return;
				job = queue.front();
				queue.pop_front();
			}
			const std::string path = prefix + std::to_string(job.index) + ".png";
			std::vector<uint8_t> png;
			const bool ok = hmrm::encode_png((int32_t)W, (int32_t)H, 4, job.pixels, W * 4, &png) &&
			                hmrm::write_file(path.c_str(), png.data(), png.size());
			hmrm_render_release(scenes[job.scene], job.ticket);
			{
				std::lock_guard<std::mutex> lk(mu);
				++free_slots[(size_t)job.scene];
			}
			cv_slot.notify_all();
			if (!ok) {
				failures.fetch_add(1);
				std::fprintf(stderr, "Failed to write screenshot to %s\n", path.c_str()); // hmap.cpp:162-164
			} else if (verbose) {
				std::printf("Saved screenshot at %s\n", path.c_str()); // hmap.cpp:165-167
			}
		}
	};

	// scene i renders frames i, i + n, i + 2n, ... (hmrm_orbit_frame_owner)
	auto producer = [&](int i) {
		std::deque<std::pair<int, int32_t>> in_flight; // (frame, ticket), oldest first
		int next = i;
		auto hand_over_oldest = [&]() -> bool {
			const std::pair<int, int32_t> fr = in_flight.front();
			in_flight.pop_front();
			const uint8_t *pixels = nullptr;
			const int rc = hmrm_render_wait(scenes[i], fr.second, &pixels, nullptr);
			if (rc != HMRM_OK && rc != HMRM_E_NOTERM) {
				note_error(rc);
				// the ticket left the in-flight list: give its ring slot back here, nobody else will
				hmrm_render_release(scenes[i], fr.second);
				{
					std::lock_guard<std::mutex> lk(mu);
					++free_slots[(size_t)i];
				}
				return false;
			}
			{
				std::lock_guard<std::mutex> lk(mu);
				queue.push_back(Job{fr.first, i, fr.second, pixels});
			}
			cv_job.notify_one();
			return true;
		};
		bool ok = true;
		while (ok && first_error.load() == HMRM_OK && (next < frames || !in_flight.empty())) {
			bool began = false;
			if (next < frames) {
				bool slot = false;
				{
					std::unique_lock<std::mutex> lk(mu);
					// with frames of its own still to hand over, do not block on a slot: go and wait for those
					if (in_flight.empty()) cv_slot.wait(lk, [&] { return free_slots[(size_t)i] > 0; });
					if (free_slots[(size_t)i] > 0) {
						--free_slots[(size_t)i];
						slot = true;
					}
				}
				if (slot) {
					hmrm_camera cam;
					hmrm_orbit_camera(base, centre_x, centre_y, radius, hang0, next, frames, &cam);
					int32_t ticket = -1;
					const int rc = hmrm_render_begin(scenes[i], &cam, &ticket);
					if (rc != HMRM_OK) {
						note_error(rc);
						break;
					}
					in_flight.emplace_back(next, ticket);
					next += n;
					began = true;
				}
			}
			// keep two frames in flight (kernel k+1 beside copy k); beyond that, or when no slot was
			// free, pass the oldest one on
			if (!in_flight.empty() && (!began || in_flight.size() > 2 || next >= frames)) ok = hand_over_oldest();
		}
		// on an error: the frames begun must still be waited for and released
		while (!in_flight.empty()) {
			const uint8_t *pixels = nullptr;
			(void)hmrm_render_wait(scenes[i], in_flight.front().second, &pixels, nullptr);
			hmrm_render_release(scenes[i], in_flight.front().second);
			in_flight.pop_front();
		}
		{
			std::lock_guard<std::mutex> lk(mu);
			--producers_left;
		}
		cv_job.notify_all();
	};

	std::vector<std::thread> pool;
	for (unsigned t = 0; t < nthreads; ++t) pool.emplace_back(encoder);
	std::vector<std::thread> renderers;
	for (int i = 0; i < n; ++i) renderers.emplace_back(producer, i);
	for (auto &t : renderers) t.join();
	for (auto &t : pool) t.join();

	int rc = first_error.load();
	if (rc != HMRM_OK) return hmrm::set_error(rc, first_error_text.c_str());
	if (failures.load() > 0) return hmrm::set_error(HMRM_E_IO, "Failed to write one or more recording frames");
	if (verbose) std::printf("Done recording.\n"); // hmap.cpp:1142
	return HMRM_OK;
}

int hmrm_record_orbit(const hmrm_scene *scene, const hmrm_camera *base, double centre_x, double centre_y,
                      double radius, double hang0, int32_t frames, const char *dir, long long id,
                      int32_t encoder_threads, int32_t verbose) {
	hmrm_scene *one = const_cast<hmrm_scene *>(scene);
	return hmrm_record_orbit_multi(&one, 1, base, centre_x, centre_y, radius, hang0, frames, dir, id, encoder_threads,
	                               verbose);
}

} // extern "C"
