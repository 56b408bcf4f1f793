// config.cpp -- whitespace-token config grammar of the reference
// (main/hmap.cpp:309-520).  The stream is consumed with the same
// `input >> value` extractions, so malformed numbers behave as there: the
// stream's failbit ends the token loop and validation runs on what was read.
// Every option is echoed (Print* functions, :197-302) into `log`; unknown keys
// give the reference's WARNING line (:487) in `warn`.
//
// Additive keys, not in the reference (it switches projection with the 1/2/3
// keys at run time, :851-868, and only ever writes PNG):
//   projection perspective|spherical|orthographic   (or 1|2|3)
//   output <path>                                   (.png or .ppm)
//   record orbit|off, sampling nearest|bilinear
#include "config.hpp"

#include <cmath>

namespace hmrm {

namespace {
// main/hmap.cpp:131-137
double degrees_to_rads(double degrees) { return (degrees / 180.0) * M_PI; }
double rads_to_degrees(double rads) { return (rads / M_PI) * 180.0; }
} // namespace

Config::Config() {
	hfov = M_PI / 2.0;  // :35
	hang = -M_PI / 4.0; // :80
	vang = M_PI / 2.0;  // :85
}

bool Config::consume(std::istream &input, std::string *fatal) {
	auto die = [&](const std::string &msg) {
		warn << msg << "\n";
		if (fatal) *fatal = msg;
		return false;
	};
	auto print_lum = [&]() { log << "lum " << lum_r << " " << lum_g << " " << lum_b << "\n"; };

	std::string next;
	while (input >> next) {
		if (next == "heightmap") {
			input >> heightmap_path;
			std::string err;
			Image img;
			if (!load_image_file(heightmap_path.c_str(), 3, &img, &err)) {
				have_heightmap = false;
				return die("Failed to load image for heightmap from " + heightmap_path + " (" + err + ")");
			}
			heightmap = std::move(img);
			have_heightmap = true;
			heightmap_dirty = true;
			log << "heightmap " << heightmap_path << "\n";
		} else if (next == "colormap") {
			input >> colormap_path;
			std::string err;
			Image img;
			if (!load_image_file(colormap_path.c_str(), 4, &img, &err)) {
				have_colormap = false;
				return die("Failed to load image for colormap from " + colormap_path + " (" + err + ")");
			}
			colormap = std::move(img);
			have_colormap = true;
			log << "colormap " << colormap_path << "\n";
		} else if (next == "print") {
			// PrintAllOptions, :282-302
			log << "print\n";
			log << "heightmap " << heightmap_path << "\n";
			log << "colormap " << colormap_path << "\n";
			log << "resolution " << screen_width << " " << screen_height << "\n";
			log << "hfov " << rads_to_degrees(hfov) << "\n";
			log << "hang " << rads_to_degrees(hang) << "\n";
			log << "vang " << rads_to_degrees(vang) << "\n";
			log << "pos " << cam_pos[0] << " " << cam_pos[1] << " " << cam_pos[2] << "\n";
			log << "min_height " << min_height << "\n";
			log << "max_height " << max_height << "\n";
			print_lum();
			log << "grid_width " << grid_width << "\n";
			log << "ortho_width " << ortho_width << "\n";
			log << "step_dist " << step_dist << "\n";
			log << "bg_color " << (int)bg_r << " " << (int)bg_g << " " << (int)bg_b << "\n";
			log << "cycle " << cycle_period << "\n";
			log << "mouse_sens " << mouse_sens << "\n";
			log << "scroll_sens " << scroll_sens << "\n";
			log << "move " << move_speed << "\n";
			log << "recording_frame_count " << recording_frame_count << "\n";
		} else if (next == "resolution") {
			input >> screen_width >> screen_height;
			log << "resolution " << screen_width << " " << screen_height << "\n";
		} else if (next == "hfov") {
			double deg;
			input >> deg;
			hfov = degrees_to_rads(deg);
			log << "hfov " << rads_to_degrees(hfov) << "\n";
		} else if (next == "hang") {
			double deg;
			input >> deg;
			hang = degrees_to_rads(deg);
			log << "hang " << rads_to_degrees(hang) << "\n";
		} else if (next == "vang") {
			double deg;
			input >> deg;
			vang = degrees_to_rads(deg);
			log << "vang " << rads_to_degrees(vang) << "\n";
		} else if (next == "pos") {
			input >> cam_pos[0] >> cam_pos[1] >> cam_pos[2];
			log << "pos " << cam_pos[0] << " " << cam_pos[1] << " " << cam_pos[2] << "\n";
		} else if (next == "pos_x") {
			input >> cam_pos[0];
			log << "pos_x " << cam_pos[0] << "\n";
		} else if (next == "pos_y") {
			input >> cam_pos[1];
			log << "pos_y " << cam_pos[1] << "\n";
		} else if (next == "pos_z") {
			input >> cam_pos[2];
			log << "pos_z " << cam_pos[2] << "\n";
		} else if (next == "min_height") {
			input >> min_height;
			heightmap_dirty = true;
			log << "min_height " << min_height << "\n";
		} else if (next == "max_height") {
			input >> max_height;
			heightmap_dirty = true;
			log << "max_height " << max_height << "\n";
		} else if (next == "lum") {
			input >> lum_r >> lum_g >> lum_b;
			heightmap_dirty = true;
			print_lum();
		} else if (next == "lum_norm") {
			double r, g, b;
			input >> r >> g >> b;
			const double total = r + g + b;
			lum_r = r / total;
			lum_g = g / total;
			lum_b = b / total;
			heightmap_dirty = true;
			print_lum();
		} else if (next == "lum_r") {
			input >> lum_r;
			heightmap_dirty = true;
			log << "lum_r " << lum_r << "\n";
		} else if (next == "lum_g") {
			input >> lum_g;
			heightmap_dirty = true;
			log << "lum_g " << lum_g << "\n";
		} else if (next == "lum_b") {
			input >> lum_b;
			heightmap_dirty = true;
			log << "lum_b " << lum_b << "\n";
		} else if (next == "grid_width") {
			input >> grid_width;
			log << "grid_width " << grid_width << "\n";
		} else if (next == "ortho_width") {
			input >> ortho_width;
			log << "ortho_width " << ortho_width << "\n";
		} else if (next == "step_dist") {
			input >> step_dist;
			log << "step_dist " << step_dist << "\n";
		} else if (next == "bg_color") {
			int r, g, b; // ints, then narrowed, as :456-461
			input >> r >> g >> b;
			bg_r = (uint8_t)r;
			bg_g = (uint8_t)g;
			bg_b = (uint8_t)b;
			log << "bg_color " << (int)bg_r << " " << (int)bg_g << " " << (int)bg_b << "\n";
		} else if (next == "cycle") {
			input >> cycle_period;
			log << "cycle " << cycle_period << "\n";
		} else if (next == "mouse_sens") {
			input >> mouse_sens;
			log << "mouse_sens " << mouse_sens << "\n";
		} else if (next == "scroll_sens") {
			input >> scroll_sens;
			log << "scroll_sens " << scroll_sens << "\n";
		} else if (next == "move") {
			input >> move_speed;
			log << "move " << move_speed << "\n";
		} else if (next == "recording_frame_count") {
			input >> recording_frame_count;
			log << "recording_frame_count " << recording_frame_count << "\n";
		} else if (next == "projection") { // additive
			std::string v;
			input >> v;
			if (v == "perspective" || v == "1") image_plane = 1;
			else if (v == "spherical" || v == "2") image_plane = 2;
			else if (v == "orthographic" || v == "3") image_plane = 3;
			else warn << "WARNING: Unknown projection: " << v << "\n";
			log << "projection "
			    << (image_plane == 1 ? "perspective" : image_plane == 2 ? "spherical" : "orthographic")
			    << "\n";
		} else if (next == "output") { // additive
			input >> output_path;
			log << "output " << output_path << "\n";
		} else if (next == "sampling") { // additive: quality mode, hmap.cpp always samples the nearest cell
			std::string v;
			input >> v;
			if (v == "nearest") sampling = 0;
			else if (v == "bilinear") sampling = 1;
			else warn << "WARNING: Unknown sampling: " << v << "\n";
			log << "sampling " << (sampling == 1 ? "bilinear" : "nearest") << "\n";
		} else if (next == "heights") { // additive: float thresholds for the nearest-cell lookup (not parity)
			std::string v;
			input >> v;
			if (v == "f32") sampling = 2;
			else if (v == "f64") sampling = sampling == 2 ? 0 : sampling;
			else warn << "WARNING: Unknown heights type: " << v << "\n";
			log << "heights " << (sampling == 2 ? "f32" : "f64") << "\n";
		} else if (next == "devices") { // additive: multi-GPU recording (BASELINE config C5)
			input >> devices;
			if (devices < 0) devices = 1;
			log << "devices " << devices << "\n";
		} else if (next == "record") { // additive: programmatic animation (hmap.cpp:907-926 is a stub)
			std::string v;
			input >> v;
			if (v == "orbit") record_mode = 1;
			else if (v == "off") record_mode = 0;
			else warn << "WARNING: Unknown record mode: " << v << "\n";
			log << "record " << (record_mode == 1 ? "orbit" : "off") << "\n";
		} else {
			warn << "WARNING: Unknown identifier: " << next << "\n";
		}
	}

	// :493-515
	if (!have_heightmap) return die("Must specify heightmap in config");
	if (!have_colormap) return die("Must specify colormap in config");
	if (heightmap.w != colormap.w || heightmap.h != colormap.h) {
		std::ostringstream m;
		m << "heightmap dimensions (" << heightmap.w << "x" << heightmap.h
		  << ") must match colormap dimensions (" << colormap.w << "x" << colormap.h << ")";
		return die(m.str());
	}
	return true;
}

} // namespace hmrm
