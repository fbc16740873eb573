/* rt_flatten.h -- host scene graph -> flat host arrays (see rt_device_scene.h). */
#ifndef RT_FLATTEN_H
#define RT_FLATTEN_H

#include <string>
#include <vector>

#include "rt_scene.h"
#include "rt_device_scene.h"

struct rt_flat_scene {
    rt_dev_camera camera;
    std::vector<rt_dev_light>    lights;
    std::vector<rt_dev_sphere>   spheres;
    std::vector<rt_dev_plane>    planes;
    std::vector<rt_dev_instance> instances;
    std::vector<rt_dev_mesh>     meshes;
    std::vector<rt_dev_material> materials;
    std::vector<rt_dev_texture>  textures;
    std::vector<rt_dev_node>     nodes;
    std::vector<rt_dev_tri>      tris;
    std::vector<rt_dev_attr>     attrs;
    std::vector<float>           texels;
    uint32_t stack_cap = 2;
    float    field_of_view = 0.0f;
    std::string error;
};

/* 0 on success; <0 and out.error set otherwise */
int rt_flatten_scene(const rt_scene *scene, rt_flat_scene &out);
void rt_flatten_camera(const rt_camera *camera, rt_dev_camera &out);
float rt_camera_focal(float field_of_view, uint32_t image_height);
float rt_pixel_norm(uint32_t spp_total);

#endif
