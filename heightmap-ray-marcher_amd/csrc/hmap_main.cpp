// hmap -- headless command line front end over libhmrm's C ABI.
//
// Same argv contract as the reference (main/hmap.cpp:526-544): exactly one
// argument, the config file; "USAGE" + exit 1 otherwise; every parsed option is
// echoed to stdout; warnings/errors go to stderr; fatal conditions exit 1.
// Instead of opening an SDL window (hmap.cpp:546-649, out of scope) it renders
// ONE full frame (`cycle 1` semantics) on the GPU and saves it the way F12 does
// (hmap.cpp:828-850 -> SavePNG :157-168): to the config's `output` path if
// given (.ppm selects binary PPM), else screenshots/hmap_<epoch>.png.
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/hmrm.h"

// `devices n` (0 = all visible).  HMRM_OVERSUBSCRIBE_DEVICES=1 (tests on a one-GPU box) keeps n above the
// number of visible GPUs and deals the scenes out round-robin over the devices there are.
static int wanted_devices(hmrm_config *cfg, int *visible_out) {
	int n = hmrm_config_devices(cfg);
	const int visible = hmrm_device_count() > 0 ? hmrm_device_count() : 1;
	const char *over = std::getenv("HMRM_OVERSUBSCRIBE_DEVICES");
	if (n <= 0) n = visible;
	if (n > visible && !(over && over[0] == '1')) n = visible;
	*visible_out = visible;
	return n;
}

static bool ends_with(const std::string &s, const char *suffix) {
	const size_t n = strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int main(int argc, char *argv[]) {
	if (argc != 2) {
		std::cerr << "USAGE: hmap.exe path/to/config.txt\n";
		return 1;
	}
	hmrm_config *cfg = hmrm_config_create();
	int rc = hmrm_config_consume_file(cfg, argv[1]);
	std::cout << hmrm_config_log(cfg);
	std::cerr << hmrm_config_warnings(cfg);
	if (rc != HMRM_OK) {
		if (rc == HMRM_E_IO) std::cerr << hmrm_last_error() << "\n";
		return 1;
	}

	hmrm_camera cam;
	hmrm_config_get_camera(cfg, &cam);
	// (the library applies the same test, check_camera; here it guards the framebuffer allocation below)
	if (cam.width <= 0 || cam.height <= 0 || (long long)cam.width * cam.height > (1LL << 31) / 4) {
		std::cerr << "resolution must be positive and at most 2^29 pixels (the reference indexes the framebuffer with int)\n";
		return 1;
	}
	hmrm_scene *scene = NULL;
	if (hmrm_config_create_scene(cfg, &scene) != HMRM_OK) {
		std::cerr << hmrm_last_error() << "\n";
		return 1;
	}

	if (hmrm_config_record_mode(cfg) == 1) {
		// `record orbit`: recording_frame_count frames on a circle around the map centre through
		// the configured camera position, always looking at the centre;
		// files screenshots/hmap_<epoch>_<n>.png as hmap.cpp:1131-1144.
		hmrm_scene_params sp;
		hmrm_config_get_scene_params(cfg, &sp);
		int32_t mw = 0, mh = 0;
		hmrm_config_height_rgb(cfg, &mw, &mh);
		const double cx = mw * sp.grid_width / 2.0, cy = -(mh * sp.grid_width) / 2.0;
		const double dx = cx - cam.pos[0], dy = cy - cam.pos[1];
		const double radius = std::sqrt(dx * dx + dy * dy);
		const double hang0 = std::atan2(dy, dx);
		std::time_t id = std::time(NULL);
		if (id == (std::time_t)(-1)) {
			std::cerr << "Failed to get time for recording. Recording NOT started.\n"; // hmap.cpp:886-892
			return 1;
		}
		std::string dir = hmrm_config_output_path(cfg);
		if (dir.empty()) dir = "screenshots";
		mkdir(dir.c_str(), 0777);
		// `devices n`: one scene per GPU, frame k on device k mod n (BASELINE config C5); the scene
		// created above lives on device 0
		int visible = 1;
		const int ndev = wanted_devices(cfg, &visible);
		std::vector<hmrm_scene *> scenes(1, scene);
		for (int d = 1; d < ndev && rc == HMRM_OK; ++d) {
			hmrm_scene *extra = NULL;
			rc = hmrm_set_device(d % visible);
			if (rc == HMRM_OK) rc = hmrm_config_create_scene(cfg, &extra);
			if (rc == HMRM_OK) scenes.push_back(extra);
		}
		if (rc == HMRM_OK)
			rc = hmrm_record_orbit_multi(scenes.data(), (int32_t)scenes.size(), &cam, cx, cy, radius, hang0,
			                             hmrm_config_recording_frame_count(cfg), dir.c_str(), (long long)id, 0, 1);
		if (rc != HMRM_OK) std::cerr << hmrm_last_error() << "\n";
		for (size_t i = 1; i < scenes.size(); ++i) hmrm_scene_destroy(scenes[i]);
		hmrm_scene_destroy(scene);
		hmrm_config_destroy(cfg);
		return rc == HMRM_OK ? 0 : 1;
	}

	std::vector<uint8_t> framebuf((size_t)cam.width * cam.height * 4);
	int visible_dev = 1;
	const int want_dev = wanted_devices(cfg, &visible_dev);
	if (want_dev > 1) {
		// `devices n`: the frame's 16-row bands are dealt out over n GPUs (BASELINE config C4)
		std::vector<hmrm_scene *> scenes(1, scene);
		for (int d = 1; d < want_dev && rc == HMRM_OK; ++d) {
			hmrm_scene *extra = NULL;
			rc = hmrm_set_device(d % visible_dev);
			if (rc == HMRM_OK) rc = hmrm_config_create_scene(cfg, &extra);
			if (rc == HMRM_OK) scenes.push_back(extra);
		}
		if (rc == HMRM_OK) rc = hmrm_render_multi(scenes.data(), (int32_t)scenes.size(), &cam, framebuf.data(), (size_t)cam.width * 4);
		for (size_t i = 1; i < scenes.size(); ++i) hmrm_scene_destroy(scenes[i]);
		if (rc != HMRM_OK && rc != HMRM_E_NOTERM) {
			std::cerr << hmrm_last_error() << "\n";
			return 1;
		}
		if (rc == HMRM_E_NOTERM) std::cerr << "WARNING: " << hmrm_last_error() << "\n";
		std::cout << "rendered " << (long long)cam.width * cam.height << " rays on " << want_dev << " devices\n";
	} else {
		hmrm_stats stats;
		rc = hmrm_render_stats(scene, &cam, framebuf.data(), (size_t)cam.width * 4, &stats, NULL, NULL);
		if (rc != HMRM_OK && rc != HMRM_E_NOTERM) {
			std::cerr << hmrm_last_error() << "\n";
			return 1;
		}
		if (rc == HMRM_E_NOTERM) std::cerr << "WARNING: " << hmrm_last_error() << "\n";
		std::cout << "rendered " << stats.rays << " rays, " << stats.steps << " ray-steps, " << stats.hits
		          << " hits in " << hmrm_last_kernel_ms() << " ms (kernel)\n";
	}

	std::string path = hmrm_config_output_path(cfg);
	if (path.empty()) {
		std::time_t seconds = std::time(NULL);
		if (seconds == (std::time_t)(-1)) {
			std::cerr << "Failed to get time for screenshot. Screenshot NOT saved.\n";
			return 1;
		}
		mkdir("screenshots", 0777);
		std::stringstream ss;
		ss << "screenshots/hmap_" << seconds << ".png";
		path = ss.str();
	}
	int wrc;
	if (ends_with(path, ".ppm") || ends_with(path, ".pnm"))
		wrc = hmrm_write_ppm(path.c_str(), cam.width, cam.height, 4, framebuf.data(), (size_t)cam.width * 4);
	else
		wrc = hmrm_write_png(path.c_str(), cam.width, cam.height, 4, framebuf.data(), (size_t)cam.width * 4);
	if (wrc != HMRM_OK)
		std::cerr << "Failed to write screenshot to " << path << "\n";
	else
		std::cout << "Saved screenshot at " << path << "\n";

	hmrm_scene_destroy(scene);
	hmrm_config_destroy(cfg);
	return wrc == HMRM_OK ? 0 : 1;
}
