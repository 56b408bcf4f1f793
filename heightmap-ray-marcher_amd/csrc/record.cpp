// record.cpp -- recording driver: a programmatic camera sweep rendered frame by
// frame and saved as PNGs, the headless counterpart of the reference's recording
// mode (main/hmap.cpp:642-649, 869-900, 1131-1144).  The reference leaves the
// per-frame parameter change to the user ("For a programmatic animation, alter
// this block and recompile", hmap.cpp:907-926, an empty stub); the sweep built
// here is the orbit of BASELINE config C5.  File naming follows
// hmap.cpp:1132-1134: <dir>/hmap_<id>_<n>.png, n = 0 .. frames-1.
//
// The GPU renders a 4K frame in well under a millisecond while the (stb-identical)
// PNG encoder needs on the order of a second per frame on one core, so frames
// are handed to a pool of encoder threads; the render loop only blocks when
// every encoder is busy (bounded memory: one frame per worker).
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

int hmrm_record_orbit(const hmrm_scene *scene, const hmrm_camera *base, double centre_x, double centre_y,
                      double radius, double hang0, int32_t frames, const char *dir, long long id,
                      int32_t encoder_threads, int32_t verbose) {
	if (!scene || !base || !dir || frames <= 0) return hmrm::set_error(HMRM_E_ARG, "hmrm_record_orbit: bad argument");
	const size_t W = (size_t)base->width, H = (size_t)base->height;
	unsigned nthreads = encoder_threads > 0 ? (unsigned)encoder_threads : std::thread::hardware_concurrency();
	if (nthreads == 0) nthreads = 4;
	if (nthreads > (unsigned)frames) nthreads = (unsigned)frames;

	struct Job {
		int index;
		std::vector<uint8_t> pixels;
	};
	std::mutex mu;
	std::condition_variable cv_job, cv_space;
	std::deque<Job> queue;
	bool closing = false;
	std::atomic<int> failures(0);
	const std::string prefix = std::string(dir) + "/hmap_" + std::to_string(id) + "_";

	auto worker = [&]() {
		for (;;) {
			Job job;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv_job.wait(lk, [&] { return closing || !queue.empty(); });
				if (queue.empty()) return;
				job = std::move(queue.front());
				queue.pop_front();
			}
			cv_space.notify_one();
			const std::string path = prefix + std::to_string(job.index) + ".png";
			std::vector<uint8_t> png;
			bool ok = hmrm::encode_png((int32_t)W, (int32_t)H, 4, job.pixels.data(), W * 4, &png) &&
			          hmrm::write_file(path.c_str(), png.data(), png.size());
			if (!ok) {
				failures.fetch_add(1);
				std::fprintf(stderr, "Failed to write screenshot to %s\n", path.c_str()); // hmap.cpp:162-164
			} else if (verbose) {
				std::printf("Saved screenshot at %s\n", path.c_str()); // hmap.cpp:165-167
			}
		}
	};
	std::vector<std::thread> pool;
	for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(worker);

	int rc = HMRM_OK;
	for (int k = 0; k < frames; ++k) {
		hmrm_camera cam;
		hmrm_orbit_camera(base, centre_x, centre_y, radius, hang0, k, frames, &cam);
		Job job;
		job.index = k;
		job.pixels.resize(W * H * 4);
		int r = hmrm_render(scene, &cam, job.pixels.data(), W * 4);
		if (r != HMRM_OK && r != HMRM_E_NOTERM) {
			rc = r;
			break;
		}
		{
			std::unique_lock<std::mutex> lk(mu);
			cv_space.wait(lk, [&] { return queue.size() < nthreads; });
			queue.push_back(std::move(job));
		}
		cv_job.notify_one();
	}
	{
		std::lock_guard<std::mutex> lk(mu);
		closing = true;
	}
	cv_job.notify_all();
	for (auto &t : pool) t.join();
	if (rc == HMRM_OK && failures.load() > 0) rc = hmrm::set_error(HMRM_E_IO, "Failed to write one or more recording frames");
	if (rc == HMRM_OK && verbose) std::printf("Done recording.\n"); // hmap.cpp:1142
	return rc;
}

} // extern "C"
