// Drop-in check: the reference's own entry file (RayTracing/RayTracing.cpp) is compiled UNCHANGED against this
// repository's host headers (its main() renamed on the command line), its scene functions are run, and the scenes they
// build are flattened and written out so the test can compare them with the scenes of raytracing_amd/host/scenes.cpp.
#include <cstdio>
#include <cstring>
#include "rt_host.h"

void Cornellbox(shared_ptr<Scene> &scene, Transform &lookat);
void sphere1(shared_ptr<Scene> &scene, Transform &lookat);
void DamagedHelmet(shared_ptr<Scene> &scene, Transform &lookat);

static void dump(const char *path, const Scene &scene, const Transform &lookat, int w, int h) {
    std::unique_ptr<FlatScene> fs = FlattenScene(scene);
    FILE *f = fopen(path, "wb");
    auto put = [&](const void *p, size_t n) { fwrite(p, 1, n, f); };
    int32_t hdr[4] = {fs->desc.n_prims, fs->desc.n_tris, fs->desc.n_spheres, fs->desc.n_nodes};
    put(hdr, sizeof(hdr));
    put(fs->nodes.data(), fs->nodes.size() * sizeof(rt_bvh_node));
    put(fs->prim_order.data(), fs->prim_order.size() * 4);
    put(fs->tri_p.data(), fs->tri_p.size() * 4);
    put(fs->tri_uv.data(), fs->tri_uv.size() * 4);
    put(fs->spheres.data(), fs->spheres.size() * sizeof(rt_sphere));
    put(fs->prim_material.data(), fs->prim_material.size() * 4);
    put(fs->materials.data(), fs->materials.size() * sizeof(rt_material));
    for (const rt_texture &t : fs->textures) { put(&t.kind, 4); put(t.value, 12); put(&t.width, 4); put(&t.height, 4); }
    put(fs->lights.data(), fs->lights.size() * sizeof(rt_light));
    Transform c2w = Inverse(lookat);
    std::unique_ptr<PerspectiveCamera> cam(CreatePerspectiveCamera(c2w, 0.01f, 40.0f, w, h));
    rt_camera rc; cam->Describe(&rc); put(&rc, sizeof(rc));
    fclose(f);
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    std::shared_ptr<Scene> scene; Transform lookat;
    if (!strcmp(argv[1], "cornell")) Cornellbox(scene, lookat);
    else if (!strcmp(argv[1], "sphere1")) sphere1(scene, lookat);
    else if (!strcmp(argv[1], "helmet")) DamagedHelmet(scene, lookat);
    else return 1;
    dump(argv[2], *scene, lookat, 800, 800);
    return 0;
}
