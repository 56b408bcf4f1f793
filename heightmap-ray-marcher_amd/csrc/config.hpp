// config.hpp -- the reference's config-file interface (main/hmap.cpp:28-112
// globals + :309-520 ConsumeConfigStream), as an object instead of globals.
#pragma once
#include <cstdint>
#include <istream>
#include <sstream>
#include <string>

#include "image_io.hpp"

namespace hmrm {

struct Config {
	// defaults = the initialisers at main/hmap.cpp:31-112
	int screen_width = 800, screen_height = 600;
	double hfov; // M_PI / 2.0, set in the constructor with hang (-M_PI/4) and vang (M_PI/2)
	double min_height = 0.0, max_height = 10.0;
	double lum_r = 0.299, lum_g = 0.587, lum_b = 0.114;
	std::string heightmap_path, colormap_path;
	Image heightmap; // RGB8  (stbi_load req_comp 3)
	Image colormap;  // RGBA8 (stbi_load req_comp 4)
	bool have_heightmap = false, have_colormap = false;
	double grid_width = 0.05;
	double step_dist = 5.0 * 0.05;
	int cycle_period = 47;
	double cam_pos[3] = {-5.0, 5.0, 0.0};
	double hang, vang;
	double mouse_sens = 1.0, scroll_sens = 1.0, move_speed = 0.05;
	double ortho_width = 2.0 * 0.05;
	int recording_frame_count = 200;
	int image_plane = 1; // IMAGEPLANE_PERSPECTIVE
	uint8_t bg_r = 0, bg_g = 0, bg_b = 0;
	// additive (north_star): output file; empty = caller decides
	std::string output_path;
	// additive: `record orbit` renders recording_frame_count frames of an orbit sweep
	int record_mode = 0;
	// additive: `devices n` -- GPUs the recording is sharded over, frame k on device k mod n (0 = all visible)
	int devices = 1;
	// additive: `sampling nearest|bilinear` (nearest = the reference's truncating lookup), `heights f32` (= 2)
	int sampling = 0;

	bool heightmap_dirty = false; // should_update_heightmap, sticky until taken
	std::ostringstream log;       // what the reference prints to stdout
	std::ostringstream warn;      // what the reference prints to stderr

	Config();

	// Returns true on success; on a condition where the reference calls
	// std::exit(1) returns false with the stderr text in *fatal (also appended
	// to `warn`).
	bool consume(std::istream &input, std::string *fatal);

};

} // namespace hmrm
