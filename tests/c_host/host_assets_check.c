/*
 * host_assets_check.c -- the host asset code (include/rt_host.h) driven from plain C and meant to be compiled with
 * -fsanitize=address,undefined (tests/test_host_sanitizers.py builds it together with rt_host_assets.c and runs it, CPU only):
 * a mesh of quads and triangles from parser-style arrays, both tree builders, census and dump, a round trip through the
 * prepared-mesh file, truncated and damaged files, a texture, the quantiser, and the error paths of bad input.
 * Prints "ok" and exits 0; any sanitizer report or failed check ends it with another status.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_host.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "check failed, line %d: %s\n", __LINE__, #cond); exit(2); } } while (0)

static uint32_t next_random(uint32_t *state) { *state = *state * 1664525u + 1013904223u; return *state >> 8; }

int main(int argc, char **argv)
{
    const char *path = argc > 1 ? argv[1] : "/tmp/rt_host_assets_check.mesh";
    /* a bumpy grid of quads (the parser's entry 0 of every array is a dummy), every fifth face a triangle */
    enum { SIDE = 40 };
    const uint32_t position_count = (SIDE + 1) * (SIDE + 1) + 1, texcoord_count = 5, normal_count = 3;
    float *positions = calloc(position_count * 3, sizeof(float));
    float texcoords[5 * 2] = {0, 0, 0, 0, 1, 0, 1, 1, 0, 1}, normals[3 * 3] = {0, 0, 0, 0, 0, 1, 0, 1, 0};
    CHECK(positions);
    uint32_t state = 12345u;
    for (uint32_t y = 0; y <= SIDE; y++)
        for (uint32_t x = 0; x <= SIDE; x++) {
            float *p = positions + 3 * (1 + y * (SIDE + 1) + x);
            p[0] = (float)x * 0.5f; p[1] = (float)y * 0.5f; p[2] = (float)(next_random(&state) % 1000u) * 1e-3f;
        }
    const uint32_t face_count = SIDE * SIDE;
    uint32_t *face_vertices = malloc(face_count * sizeof(uint32_t));
    uint32_t *indices = malloc((size_t)face_count * 4 * 3 * sizeof(uint32_t));
    CHECK(face_vertices && indices);
    uint64_t index_count = 0;
    for (uint32_t y = 0; y < SIDE; y++)
        for (uint32_t x = 0; x < SIDE; x++) {
            const uint32_t f = y * SIDE + x, corners = f % 5u == 4u ? 3u : 4u;
            const uint32_t a = 1 + y * (SIDE + 1) + x, quad[4] = {a, a + 1, a + SIDE + 2, a + SIDE + 1};
            face_vertices[f] = corners;
            for (uint32_t k = 0; k < corners; k++) {
                indices[3 * index_count] = quad[k]; indices[3 * index_count + 1] = 1 + k; indices[3 * index_count + 2] = 1 + (f & 1u);
                index_count++;
            }
        }

    rt_mesh mesh;
    memset(&mesh, 0, sizeof(mesh));
    CHECK(rt_host_mesh_from_obj_arrays(positions, position_count, texcoords, texcoord_count, normals, normal_count,
                                       indices, index_count, face_vertices, face_count, &mesh) == 0);
    uint32_t census[4];
    rt_host_mesh_census(&mesh, census);
    CHECK(census[0] > 100 && census[1] > 50 && census[2] > 5 && census[3] >= 1 && census[3] <= 16);
    uint32_t *dump = malloc((size_t)census[0] * 8 * sizeof(uint32_t));
    CHECK(dump && rt_host_mesh_dump_tree(&mesh, dump) == census[0]);

    /* the prepared-mesh file: round trip, wrong key, truncated, damaged */
    CHECK(rt_host_mesh_save(&mesh, 0xC0FFEEull, path) == 0);
    rt_mesh again;
    memset(&again, 0, sizeof(again));
    CHECK(rt_host_mesh_load(path, 0xC0FFEEull, &again) == 0);
    uint32_t census_again[4];
    rt_host_mesh_census(&again, census_again);
    CHECK(memcmp(census, census_again, sizeof(census)) == 0);
    uint32_t *dump_again = malloc((size_t)census[0] * 8 * sizeof(uint32_t));
    CHECK(dump_again && rt_host_mesh_dump_tree(&again, dump_again) == census[0]);
    CHECK(memcmp(dump, dump_again, (size_t)census[0] * 8 * sizeof(uint32_t)) == 0);
    rt_host_mesh_free(&again);
    memset(&again, 0, sizeof(again));
    CHECK(rt_host_mesh_load(path, 0xBADull, &again) == -7);
    CHECK(rt_host_mesh_load("/nonexistent/dir/file.mesh", 0xC0FFEEull, &again) == -3);
    {
        FILE *f = fopen(path, "rb");
        CHECK(f);
        CHECK(fseek(f, 0, SEEK_END) == 0);
        const long size = ftell(f);
        CHECK(size > 1000 && fseek(f, 0, SEEK_SET) == 0);
        unsigned char *bytes = malloc((size_t)size);
        CHECK(bytes && fread(bytes, 1, (size_t)size, f) == (size_t)size);
        fclose(f);
        char other[1024];
        snprintf(other, sizeof(other), "%s.damaged", path);
        const long cuts[4] = {0, 17, size / 2, size - 1};
        for (int c = 0; c < 4; c++) {                                /* truncated at several places */
            f = fopen(other, "wb");
            CHECK(f && fwrite(bytes, 1, (size_t)cuts[c], f) == (size_t)cuts[c]);
            fclose(f);
            memset(&again, 0, sizeof(again));
            CHECK(rt_host_mesh_load(other, 0xC0FFEEull, &again) < 0);
        }
        for (int trial = 0; trial < 200; trial++) {                  /* a random word overwritten: refused, or loaded without a bad access */
            unsigned char *copy = malloc((size_t)size);
            CHECK(copy);
            memcpy(copy, bytes, (size_t)size);
            const size_t at = (next_random(&state) % (uint32_t)(size / 4)) * 4u;
            const uint32_t junk = trial % 3 == 0 ? 0xFFFFFFFFu : next_random(&state) * 2654435761u;
            memcpy(copy + at, &junk, 4);
            f = fopen(other, "wb");
            CHECK(f && fwrite(copy, 1, (size_t)size, f) == (size_t)size);
            fclose(f);
            free(copy);
            memset(&again, 0, sizeof(again));
            if (rt_host_mesh_load(other, 0xC0FFEEull, &again) == 0) {
                uint32_t c2[4];
                rt_host_mesh_census(&again, c2);
                rt_host_mesh_free(&again);
            }
        }
        remove(other);
        free(bytes);
    }

    /* the second builder on the same mesh, then the first one again */
    CHECK(rt_host_mesh_build_tree_sah(&mesh, 4) == 0);
    uint32_t census_sah[4];
    rt_host_mesh_census(&mesh, census_sah);
    CHECK(census_sah[3] <= 16 && census_sah[1] >= census[1]);
    CHECK(rt_host_mesh_build_tree(&mesh) == 0);
    rt_host_mesh_census(&mesh, census_again);
    CHECK(census_again[3] <= 16);
    rt_host_mesh_free(&mesh);

    /* bad input: a face that names more corners than there are indices, an index beyond the positions */
    memset(&mesh, 0, sizeof(mesh));
    CHECK(rt_host_mesh_from_obj_arrays(positions, position_count, texcoords, texcoord_count, normals, normal_count,
                                       indices, index_count - 2, face_vertices, face_count, &mesh) == -6);
    indices[3 * 7] = position_count + 5;
    memset(&mesh, 0, sizeof(mesh));
    CHECK(rt_host_mesh_from_obj_arrays(positions, position_count, texcoords, texcoord_count, normals, normal_count,
                                       indices, index_count, face_vertices, face_count, &mesh) < 0);

    /* a texture and the quantiser */
    enum { TW = 37, TH = 19 };
    uint8_t *rgb = malloc(TW * TH * 3), *back = malloc(TW * TH * 3);
    CHECK(rgb && back);
    for (int i = 0; i < TW * TH * 3; i++) rgb[i] = (uint8_t)(next_random(&state) & 255u);
    rt_image3f texture;
    memset(&texture, 0, sizeof(texture));
    CHECK(rt_host_texture_from_rgb8(rgb, TW, TH, &texture) == 0);
    CHECK(texture.size.x == TW && texture.size.y == TH && texture.pixels);
    rt_host_image_quantize(&texture, back);
    rt_host_image_free(&texture);

    free(rgb); free(back); free(dump); free(dump_again); free(indices); free(face_vertices); free(positions);
    remove(path);
    puts("ok");
    return 0;
}
