// host/host_api.cpp -- C entry points of the headless driver library (libhiprenderer_host.so).
//
// Python (tests, bench.py) drives the C++ host code through these; they hold no rendering logic.
// The scene a handle owns is exactly what a Bifrost host would hand to hipr_upload_scene().
#include "Scenes.h"

#include <cstring>
#include <string>

using namespace HIPRenderer;

extern "C" {

// name: "cornell" | "atrium" | "quad" | "empty_ortho". variant bit 0: force every material to the Diffuse
// shading model (BASELINE.json config 2). param0/param1: atrium target triangles + seed, ortho width + height.
void* hiprh_scene_create(const char* name, unsigned variant, unsigned param0, unsigned param1) {
    if (!name) return nullptr;
    SceneBuilder* sb = new SceneBuilder();
    std::string n = name;
    if (n == "cornell") Scenes::create_cornell_box(*sb, param0 ? param0 : 1u);   // param0: quads per wall edge
    else if (n == "atrium") Scenes::create_atrium(*sb, param0 ? param0 : 260000u, param1 ? param1 : 1u);
    else if (n == "quad") Scenes::create_quad_scene(*sb, param0, param1);
    else if (n == "empty_ortho") Scenes::create_empty_ortho_scene(*sb, param0, param1, RGB(0.1f, 0.5f, 2.0f));
    else { delete sb; return nullptr; }
    if (variant & 1u) sb->force_shading_model(HIPR_SHADING_DIFFUSE);
    sb->finalize();
    return sb;
}

void hiprh_scene_destroy(void* scene) { delete static_cast<SceneBuilder*>(scene); }

const HiprSceneDesc* hiprh_scene_desc(void* scene) { return scene ? &static_cast<SceneBuilder*>(scene)->desc() : nullptr; }

int hiprh_scene_state(void* scene, HiprSceneState* out) {
    if (!scene || !out) return -1;
    *out = static_cast<SceneBuilder*>(scene)->state();
    return 0;
}

// max_bounce_count < 0 keeps the scene's own default.
int hiprh_scene_camera(void* scene, unsigned width, unsigned height, unsigned accumulations, int max_bounce_count, float pdf_scale, HiprCameraState* out) {
    if (!scene || !out || !width || !height) return -1;
    SceneBuilder* sb = static_cast<SceneBuilder*>(scene);
    CameraDescription cam = sb->camera;
    if (max_bounce_count >= 0) cam.max_bounce_count = unsigned(max_bounce_count);
    *out = make_camera_state(cam, float(width) / float(height), accumulations, pdf_scale);
    return 0;
}

// Stand-alone BVH build over caller triangles (tests): returns a handle owning nodes + order.
struct BvhHandle { BvhBuildResult result; };
void* hiprh_bvh_build(const HiprTriangle* triangles, unsigned count, unsigned max_depth) {
    std::vector<HiprTriangle> t(triangles, triangles + count);
    BvhHandle* h = new BvhHandle();
    h->result = build_bvh(t, max_depth);
    return h;
}
unsigned hiprh_bvh_node_count(void* h) { return unsigned(static_cast<BvhHandle*>(h)->result.nodes.size()); }
unsigned hiprh_bvh_max_depth(void* h) { return static_cast<BvhHandle*>(h)->result.max_depth; }
const HiprBvhNode* hiprh_bvh_nodes(void* h) { return static_cast<BvhHandle*>(h)->result.nodes.data(); }
const unsigned* hiprh_bvh_order(void* h) { return static_cast<BvhHandle*>(h)->result.order.data(); }
unsigned hiprh_bvh_wide_node_count(void* h) { return unsigned(static_cast<BvhHandle*>(h)->result.wide_nodes.size()); }
unsigned hiprh_bvh_wide_stack_entries(void* h) { return static_cast<BvhHandle*>(h)->result.wide_stack_entries; }
const HiprWideNode* hiprh_bvh_wide_nodes(void* h) { return static_cast<BvhHandle*>(h)->result.wide_nodes.data(); }
void hiprh_bvh_destroy(void* h) { delete static_cast<BvhHandle*>(h); }

void hiprh_encode_octahedral(const float* normals_n3, int n, short* out_n2) {
    for (int i = 0; i < n; ++i) {
        Bifrost::Math::OctahedralNormal e = Bifrost::Math::OctahedralNormal::encode_precise({normals_n3[3 * i], normals_n3[3 * i + 1], normals_n3[3 * i + 2]});
        out_n2[2 * i] = e.encoding.x;
        out_n2[2 * i + 1] = e.encoding.y;
    }
}

} // extern "C"
