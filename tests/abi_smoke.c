/* abi_smoke.c -- include/hmrm.h is a C header: this C99 program links libhmrm.so and drives the entry points
 * that need no GPU (image IO, config grammar, host-side frame record).  Built and run by tests/test_abi.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hmrm.h"

#define CHECK(cond)                                                          \
	do {                                                                     \
		if (!(cond)) {                                                       \
			fprintf(stderr, "abi_smoke: %s failed (line %d): %s\n", #cond, __LINE__, hmrm_last_error()); \
			return 1;                                                        \
		}                                                                    \
	} while (0)

int main(int argc, char **argv) {
	if (argc != 2) return 2;
	const char *dir = argv[1];
	char hpath[512], cpath[512], text[2048];
	uint8_t rgb[8 * 6 * 3], rgba[8 * 6 * 4];
	int i;
	for (i = 0; i < 8 * 6; ++i) {
		rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = (uint8_t)(i * 5);
		rgba[4 * i] = (uint8_t)i, rgba[4 * i + 1] = (uint8_t)(2 * i), rgba[4 * i + 2] = (uint8_t)(3 * i), rgba[4 * i + 3] = 255;
	}
	snprintf(hpath, sizeof hpath, "%s/h.ppm", dir);
	snprintf(cpath, sizeof cpath, "%s/c.png", dir);
	CHECK(hmrm_abi_version() == HMRM_ABI_VERSION);
	CHECK(hmrm_write_ppm(hpath, 8, 6, 3, rgb, 8 * 3) == HMRM_OK);
	CHECK(hmrm_write_png(cpath, 8, 6, 4, rgba, 8 * 4) == HMRM_OK);

	uint8_t *px = NULL;
	int32_t w = 0, h = 0, n = 0;
	CHECK(hmrm_image_load(cpath, 4, &px, &w, &h, &n) == HMRM_OK && w == 8 && h == 6 && n == 4);
	CHECK(memcmp(px, rgba, sizeof rgba) == 0);
	hmrm_image_free(px);

	hmrm_config *cfg = hmrm_config_create();
	CHECK(cfg != NULL);
	snprintf(text, sizeof text, "resolution 16 9 hfov 60 pos 1 2 3 step_dist 0.5 projection spherical devices 2\n"
	                            "heightmap %s colormap %s bogus_key 1\n", hpath, cpath);
	CHECK(hmrm_config_consume_string(cfg, text) == HMRM_OK);
	CHECK(strstr(hmrm_config_log(cfg), "resolution 16 9") != NULL);
	CHECK(strstr(hmrm_config_warnings(cfg), "Unknown identifier: bogus_key") != NULL);
	hmrm_camera cam;
	hmrm_scene_params sp;
	hmrm_config_get_camera(cfg, &cam);
	hmrm_config_get_scene_params(cfg, &sp);
	CHECK(cam.width == 16 && cam.height == 9 && cam.projection == HMRM_SPHERICAL && cam.sampling == HMRM_NEAREST);
	CHECK(hmrm_config_devices(cfg) == 2 && sp.grid_width == 0.05);
	const uint8_t *hm = hmrm_config_height_rgb(cfg, &w, &h);
	CHECK(hm != NULL && w == 8 && h == 6 && memcmp(hm, rgb, sizeof rgb) == 0);

	/* the per-frame record the kernel would get (host only) */
	double rec[25], tables[2 * 16 + 2 * 9];
	CHECK(hmrm_debug_frame(&cam, &sp, 8, 6, rec, tables) == HMRM_OK);
	CHECK(rec[0] == 1.0 && rec[1] == 2.0 && rec[2] == 3.0 && rec[22] == 0.5);

	/* without a GPU a scene cannot be created -- and says so; with one this works */
	hmrm_scene *scene = NULL;
	int rc = hmrm_config_create_scene(cfg, &scene);
	CHECK(rc == HMRM_OK || rc == HMRM_E_DEVICE);
	if (rc == HMRM_OK) hmrm_scene_destroy(scene);
	CHECK(hmrm_orbit_frame_owner(10, 8) == 2);
	{ /* the pyramid layout rule needs no GPU: a 4096^2 map fits the kernel's 32-bit look-up offsets, 16385 x 32766 does not */
		int32_t row = 0, shift = 0, levels = 0;
		CHECK(hmrm_debug_mip_layout(4096, 4096, &row, &shift, &levels) == 1 && row == 2048 && shift == 22 && levels == 7);
		CHECK(hmrm_debug_mip_layout(16385, 32766, &row, &shift, &levels) == 0 && shift == 28);
		CHECK(hmrm_debug_mip_layout(0, 1, NULL, NULL, NULL) < 0);
	}
	hmrm_config_destroy(cfg);
	printf("abi_smoke ok\n");
	return 0;
}
