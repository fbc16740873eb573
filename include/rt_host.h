/*
 * rt_host.h -- host-side asset preparation around the GPU render loop (plain C):
 * triangulation + bounding-box tree of a mesh, texture linearisation, and the
 * 8-bit output quantiser.  Implemented in raytracing_amd/csrc/rt_host_assets.c.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Replaces split_mesh_node (reference: src/main.c:92-175) called on the root as
 * load_mesh does (src/main.c:256-258): rebuilds mesh->root and mesh->tree_depth
 * and permutes mesh->faces in place, node for node and swap for swap as the
 * reference.  mesh->root.childs must be NULL or a tree made by one of these builders (it is freed).
 * Returns 0, or <0 on bad input / out of memory. */
int rt_host_mesh_build_tree(rt_mesh *mesh);

/* A second builder with no counterpart in the reference: per node the axis and plane of least
 * surface-area cost over 16 centroid bins; a node of more than leaf_triangles (0 = 4) is split as long as
 * a plane lowers the cost, and always when it holds more than 16.  Same
 * records as above (so the oracle, the flattener and the kernels take it unchanged), far fewer box and
 * triangle tests per ray; pictures agree with the reference tree's except where the tree shape shows
 * (flat boxes, equal distances).  Replaces the mesh's current tree. */
int rt_host_mesh_build_tree_sah(rt_mesh *mesh, uint32_t leaf_triangles);

/* Replaces the body of load_mesh after the OBJ parse (reference:
 * src/main.c:183-258): copies the parser's arrays (entry 0 of each is the
 * parser's dummy), applies the axis permutation, splits every polygon into a
 * fan of triangles (a quad gives (0,1,2),(0,2,3) as the reference) and builds
 * the tree.  indices_ptn holds index_count (position, texcoord, normal) index
 * triples, face after face; -6 if the faces name more corners than that. */
int rt_host_mesh_from_obj_arrays(const float *positions, uint32_t position_count,
                                 const float *texcoords, uint32_t texcoord_count,
                                 const float *normals, uint32_t normal_count,
                                 const uint32_t *indices_ptn, uint64_t index_count,
                                 const uint32_t *face_vertices, uint32_t face_count, rt_mesh *out);

void rt_host_mesh_free(rt_mesh *mesh);

/* A prepared mesh (what load_mesh, src/main.c:177-262, leaves in memory: attribute arrays, permuted
 * face table, tree) as one file, so that a later run skips the OBJ parse and the tree build.  `key`
 * is the caller's content key (a hash of the source model and the builder's parameters); load refuses
 * a file written under another key (-7), a missing file (-3) and a malformed one (-4, -5: every
 * index, range and count is checked).  No counterpart in the reference. */
int rt_host_mesh_save(const rt_mesh *mesh, uint64_t key, const char *path);
int rt_host_mesh_load(const char *path, uint64_t key, rt_mesh *out);

/* {nodes, leaves, deepest level (root = 1), largest leaf} */
void rt_host_mesh_census(const rt_mesh *mesh, uint32_t out[4]);

/* Depth-first dump, 8 words per node: bbox (6 float bit patterns), first
 * triangle id, last triangle id (bit 31 set on interior nodes).  The buffer
 * must hold census()[0] nodes.  Returns the number of nodes written. */
size_t rt_host_mesh_dump_tree(const rt_mesh *mesh, uint32_t *words8_per_node);

/* Replaces the conversion loop of load_texture (reference: src/main.c:52-59):
 * 8-bit RGB rows, top row first, become a linear-light float image stored
 * bottom row first. */
int rt_host_texture_from_rgb8(const uint8_t *rgb, uint32_t width, uint32_t height, rt_image3f *out);
void rt_host_image_free(rt_image3f *image);

/* Replaces the conversion loop of save_image (reference: src/main.c:73-82):
 * clamp to 1, gamma 1/2.2, scale by 255, truncate.  rgb holds 3*w*h bytes. */
void rt_host_image_quantize(const rt_image3f *image, uint8_t *rgb);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
