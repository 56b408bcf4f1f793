// config.cpp -- whitespace-token config grammar of the reference
// (main/hmap.cpp:309-520).  The stream is consumed with the same
// `input >> value` extractions, so malformed numbers behave as there: the
// stream's failbit ends the token loop and validation runs on what was read.
// Every option is echoed (Print* functions, :197-302) into `log`; unknown keys
// give the reference's WARNING line (:487) in `warn`.
//
// The grammar is data here: one table row per key -- what it reads, what it
// touches, how it is echoed -- instead of a chain of comparisons.
//
// Additive keys, not in the reference (it switches projection with the 1/2/3
// keys at run time, :851-868, and only ever writes PNG):
//   projection perspective|spherical|orthographic   (or 1|2|3)
//   output <path>                                   (.png or .ppm)
//   record orbit|off, devices n, sampling nearest|bilinear, heights f64|f32
#include "config.hpp"

#include <cmath>
#include <cstring>

namespace hmrm {

namespace {

// main/hmap.cpp:131-137
double degrees_to_rads(double degrees) { return (degrees / 180.0) * M_PI; }
double rads_to_degrees(double rads) { return (rads / M_PI) * 180.0; }

// What a key's handler may report: nothing, or the message the reference prints before std::exit(1).
using Handler = void (*)(Config &c, std::istream &in, const char *key, std::string *die);

// ---- generic rows ----
template <auto Member>
void scalar(Config &c, std::istream &in, const char *key, std::string *) {
	in >> c.*Member;
	c.log << key << " " << c.*Member << "\n";
}
// ... that also invalidate the converted heights (should_update_heightmap, :401-440)
template <auto Member>
void height_scalar(Config &c, std::istream &in, const char *key, std::string *die) {
	scalar<Member>(c, in, key, die);
	c.heightmap_dirty = true;
}
// angles are given in degrees and kept in radians; the echo converts back (:367-384)
template <auto Member>
void angle(Config &c, std::istream &in, const char *key, std::string *) {
	double deg = 0.0; // (a key at the very end of a stream extracts nothing: the reference reads an uninitialised local there)
	in >> deg;
	c.*Member = degrees_to_rads(deg);
	c.log << key << " " << rads_to_degrees(c.*Member) << "\n";
}
template <int Axis>
void position_axis(Config &c, std::istream &in, const char *key, std::string *) {
	in >> c.cam_pos[Axis];
	c.log << key << " " << c.cam_pos[Axis] << "\n";
}
// a word out of a short list -> an integer setting; an unknown word warns and leaves the setting alone
struct Word { const char *text; int value; };
template <size_t N>
bool pick(std::istream &in, const Word (&words)[N], int *out, std::string *seen) {
	in >> *seen;
	for (const Word &w : words)
		if (*seen == w.text) {
			*out = w.value;
			return true;
		}
	return false;
}

// ---- the rows that do more ----
void echo_lum(Config &c) { c.log << "lum " << c.lum_r << " " << c.lum_g << " " << c.lum_b << "\n"; }

template <int Comp>
void map_image(Config &c, std::istream &in, const char *key, std::string *die) {
	std::string &path = Comp == 3 ? c.heightmap_path : c.colormap_path;
	bool &have = Comp == 3 ? c.have_heightmap : c.have_colormap;
	in >> path;
	std::string err;
	Image img;
	if (!load_image_file(path.c_str(), Comp, &img, &err)) { // stbi_load(path, .., req_comp), :320-329 / :341-350
		have = false;
		*die = std::string("Failed to load image for ") + key + " from " + path + " (" + err + ")";
		return;
	}
	(Comp == 3 ? c.heightmap : c.colormap) = std::move(img);
	have = true;
	if (Comp == 3) c.heightmap_dirty = true;
	c.log << key << " " << path << "\n";
}

void print_all(Config &c, std::istream &, const char *key, std::string *) { // PrintAllOptions, :282-302
	c.log << key << "\n"
	      << "heightmap " << c.heightmap_path << "\n"
	      << "colormap " << c.colormap_path << "\n"
	      << "resolution " << c.screen_width << " " << c.screen_height << "\n"
	      << "hfov " << rads_to_degrees(c.hfov) << "\n"
	      << "hang " << rads_to_degrees(c.hang) << "\n"
	      << "vang " << rads_to_degrees(c.vang) << "\n"
	      << "pos " << c.cam_pos[0] << " " << c.cam_pos[1] << " " << c.cam_pos[2] << "\n"
	      << "min_height " << c.min_height << "\n"
	      << "max_height " << c.max_height << "\n";
	echo_lum(c);
	c.log << "grid_width " << c.grid_width << "\n"
	      << "ortho_width " << c.ortho_width << "\n"
	      << "step_dist " << c.step_dist << "\n"
	      << "bg_color " << (int)c.bg_r << " " << (int)c.bg_g << " " << (int)c.bg_b << "\n"
	      << "cycle " << c.cycle_period << "\n"
	      << "mouse_sens " << c.mouse_sens << "\n"
	      << "scroll_sens " << c.scroll_sens << "\n"
	      << "move " << c.move_speed << "\n"
	      << "recording_frame_count " << c.recording_frame_count << "\n";
}

void resolution(Config &c, std::istream &in, const char *key, std::string *) {
	in >> c.screen_width >> c.screen_height;
	c.log << key << " " << c.screen_width << " " << c.screen_height << "\n";
}

void position(Config &c, std::istream &in, const char *key, std::string *) {
	in >> c.cam_pos[0] >> c.cam_pos[1] >> c.cam_pos[2];
	c.log << key << " " << c.cam_pos[0] << " " << c.cam_pos[1] << " " << c.cam_pos[2] << "\n";
}

template <bool Normalise>
void lum_weights(Config &c, std::istream &in, const char *, std::string *) {
	double w[3] = {0.0, 0.0, 0.0};
	in >> w[0] >> w[1] >> w[2];
	const double total = Normalise ? w[0] + w[1] + w[2] : 1.0; // lum_norm divides by the sum (:417-427)
	c.lum_r = Normalise ? w[0] / total : w[0];
	c.lum_g = Normalise ? w[1] / total : w[1];
	c.lum_b = Normalise ? w[2] / total : w[2];
	c.heightmap_dirty = true;
	echo_lum(c);
}

void background(Config &c, std::istream &in, const char *key, std::string *) {
	int v[3] = {0, 0, 0}; // read as ints, then narrowed (:456-461)
	in >> v[0] >> v[1] >> v[2];
	c.bg_r = (uint8_t)v[0];
	c.bg_g = (uint8_t)v[1];
	c.bg_b = (uint8_t)v[2];
	c.log << key << " " << (int)c.bg_r << " " << (int)c.bg_g << " " << (int)c.bg_b << "\n";
}

// -- additive keys --
void projection(Config &c, std::istream &in, const char *key, std::string *) {
	static const Word words[] = {{"perspective", 1}, {"1", 1}, {"spherical", 2}, {"2", 2}, {"orthographic", 3}, {"3", 3}};
	static const char *const names[] = {"", "perspective", "spherical", "orthographic"};
	std::string seen;
	if (!pick(in, words, &c.image_plane, &seen)) c.warn << "WARNING: Unknown projection: " << seen << "\n";
	c.log << key << " " << names[c.image_plane] << "\n";
}

void sampling_mode(Config &c, std::istream &in, const char *key, std::string *) { // quality mode; hmap.cpp always samples the nearest cell
	static const Word words[] = {{"nearest", 0}, {"bilinear", 1}};
	std::string seen;
	if (!pick(in, words, &c.sampling, &seen)) c.warn << "WARNING: Unknown sampling: " << seen << "\n";
	c.log << key << " " << (c.sampling == 1 ? "bilinear" : "nearest") << "\n";
}

void heights_type(Config &c, std::istream &in, const char *key, std::string *) { // float thresholds for the nearest-cell lookup (not parity)
	static const Word words[] = {{"f64", 0}, {"f32", 1}};
	std::string seen;
	int f32 = 0;
	if (!pick(in, words, &f32, &seen)) c.warn << "WARNING: Unknown heights type: " << seen << "\n";
	else if (f32) c.sampling = 2;
	else if (c.sampling == 2) c.sampling = 0;
	c.log << key << " " << (c.sampling == 2 ? "f32" : "f64") << "\n";
}

void device_count_key(Config &c, std::istream &in, const char *key, std::string *) { // multi-GPU recording (BASELINE config C5)
	in >> c.devices;
	if (c.devices < 0) c.devices = 1;
	c.log << key << " " << c.devices << "\n";
}

void record_mode_key(Config &c, std::istream &in, const char *key, std::string *) { // programmatic animation (hmap.cpp:907-926 is a stub)
	static const Word words[] = {{"off", 0}, {"orbit", 1}};
	std::string seen;
	if (!pick(in, words, &c.record_mode, &seen)) c.warn << "WARNING: Unknown record mode: " << seen << "\n";
	c.log << key << " " << (c.record_mode == 1 ? "orbit" : "off") << "\n";
}

struct Row { const char *key; Handler apply; };
const Row kGrammar[] = {
	// the reference's 27 keys (main/hmap.cpp:314-488)
	{"heightmap", map_image<3>},
	{"colormap", map_image<4>},
	{"print", print_all},
	{"resolution", resolution},
	{"hfov", angle<&Config::hfov>},
	{"hang", angle<&Config::hang>},
	{"vang", angle<&Config::vang>},
	{"pos", position},
	{"pos_x", position_axis<0>},
	{"pos_y", position_axis<1>},
	{"pos_z", position_axis<2>},
	{"min_height", height_scalar<&Config::min_height>},
	{"max_height", height_scalar<&Config::max_height>},
	{"lum", lum_weights<false>},
	{"lum_norm", lum_weights<true>},
	{"lum_r", height_scalar<&Config::lum_r>},
	{"lum_g", height_scalar<&Config::lum_g>},
	{"lum_b", height_scalar<&Config::lum_b>},
	{"grid_width", scalar<&Config::grid_width>},
	{"ortho_width", scalar<&Config::ortho_width>},
	{"step_dist", scalar<&Config::step_dist>},
	{"bg_color", background},
	{"cycle", scalar<&Config::cycle_period>},
	{"mouse_sens", scalar<&Config::mouse_sens>},
	{"scroll_sens", scalar<&Config::scroll_sens>},
	{"move", scalar<&Config::move_speed>},
	{"recording_frame_count", scalar<&Config::recording_frame_count>},
	// additive
	{"projection", projection},
	{"output", scalar<&Config::output_path>},
	{"sampling", sampling_mode},
	{"heights", heights_type},
	{"devices", device_count_key},
	{"record", record_mode_key},
};

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
	std::string token;
	while (input >> token) {
		const Row *row = nullptr;
		for (const Row &r : kGrammar)
			if (token == r.key) row = &r;
		if (!row) {
			warn << "WARNING: Unknown identifier: " << token << "\n"; // :487
			continue;
		}
		std::string fatal_msg;
		row->apply(*this, input, row->key, &fatal_msg);
		if (!fatal_msg.empty()) return die(fatal_msg);
	}
	// end of stream: both maps present and of one size (:493-515)
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
