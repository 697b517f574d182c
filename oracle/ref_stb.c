/* TEST INFRASTRUCTURE — the reference's own image decoder as a checker.
 *
 * The reference vendors stb_image v2.27 (cuda/stb_image.h) and decodes every texture with
 * stbi_load(filename, &w, &h, &n, 3) (cuda/main.cu:393).  This file compiles that header, where
 * it lies in the reference checkout, into oracle/_ref/libref_stb.so (see oracle/Makefile) so that
 * tests can hold raytrace_amd's decoder against it byte for byte.  SIMD paths are switched off:
 * the header states that they produce the same bytes as the plain C paths.  Only tests/ and
 * oracle/gen_golden_jpeg.py may load the library.
 */
#define STB_IMAGE_IMPLEMENTATION
#define STBI_NO_SIMD
#define STBI_ONLY_JPEG
#include "cuda/stb_image.h"

/* RGB8 rows, top row first; NULL on failure.  Free with ref_stbi_free. */
unsigned char* ref_stbi_load_rgb(const unsigned char* bytes, int n_bytes, int* w, int* h) {
    int comp = 0;
    return stbi_load_from_memory(bytes, n_bytes, w, h, &comp, 3);
}
void ref_stbi_free(unsigned char* p) { stbi_image_free(p); }
const char* ref_stbi_reason(void) { return stbi_failure_reason(); }
