/* hiprenderer_c.h -- C-ABI of the HIPRenderer path-tracing kernels for MI355X (gfx950).
 *
 * This is the drop-in boundary between a Bifrost host renderer and the hand-written HIP
 * wavefront path tracer. Every entry point replaces one interaction the reference
 * OptiXRenderer host code has with the OptiX 6.5 runtime. Paths below are relative to
 * /root/reference/extensions/OptiXRenderer/OptiXRenderer/ ("OR/").
 *
 * Conventions
 *  - All functions return an int status: HIPR_OK (0) or a negative HiprStatus. No exceptions
 *    cross the ABI. hipr_last_error() returns a thread-local human readable message.
 *  - The library never takes ownership of caller memory: uploads copy.
 *  - Pointers named *_device are device (HBM) pointers, everything else is host memory.
 *  - Single-threaded contract per context, as the reference (BF/Core/Engine.cpp:36-49).
 *  - Image rows: row 0 is the bottom row (the reference adaptor flips Y when blitting,
 *    extensions/DX11OptiXAdapter/DX11OptiXAdapter/Adaptor.cpp:113-115).
 */
#ifndef HIPRENDERER_C_H
#define HIPRENDERER_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------- */
/* Status codes                                                                                */
/* ------------------------------------------------------------------------------------------- */
typedef enum HiprStatus {
    HIPR_OK = 0,
    HIPR_ERROR_INVALID_ARGUMENT = -1,
    HIPR_ERROR_NO_DEVICE = -2,      /* OR/Renderer.cpp:280-281 returns an invalid renderer */
    HIPR_ERROR_OUT_OF_MEMORY = -3,
    HIPR_ERROR_HIP = -4,            /* a HIP runtime call failed, see hipr_last_error() */
    HIPR_ERROR_NOT_READY = -5,      /* render before tables/scene/frame were set */
    HIPR_ERROR_UNSUPPORTED = -6,
    HIPR_ERROR_TIMEOUT = -7         /* a device group's tile exchange did not finish within its deadline (hipr_group_accumulate_samples); hipr_last_error() names the member */
} HiprStatus;

/* ------------------------------------------------------------------------------------------- */
/* Device PODs. Layouts follow the reference's device structs so the host code fills them the  */
/* same way (OR/Types.h). All are tightly packed little-endian.                                */
/* ------------------------------------------------------------------------------------------- */

/* OR/Types.h:353-383, 64 bytes. flags: 1 = ThinWalled, 2 = Cutout. shading_model: 0 Default,
 * 1 Diffuse, 2 Transmissive. coat / coat_roughness are UNorm16 (OR/Types.h:76-93). Texture IDs
 * are indices into HiprSceneDesc::textures, 0 = none. */
typedef struct HiprMaterial {
    uint16_t flags;
    uint16_t shading_model;
    float tint[3];
    float roughness;
    int32_t tint_roughness_texture_ID;
    int32_t roughness_texture_ID;
    float specularity;
    float metallic;
    int32_t metallic_texture_ID;
    float coverage;
    int32_t coverage_texture_ID;
    float emission[3];
    uint16_t coat;
    uint16_t coat_roughness;
} HiprMaterial;

enum { HIPR_MATERIAL_THIN_WALLED = 1, HIPR_MATERIAL_CUTOUT = 2 };
enum { HIPR_SHADING_DEFAULT = 0, HIPR_SHADING_DIFFUSE = 1, HIPR_SHADING_TRANSMISSIVE = 2 };

/* OR/Types.h:290-312, 48 bytes. `data` is the union payload:
 *   Sphere      : power[0..2] position[3..5] radius[6]
 *   Spot        : power[0..2] position[3..5] radius[6] direction[7..9] cos_angle[10]
 *   Directional : radiance[0..2] direction[3..5]
 * flags & 7 is the light type. */
typedef struct HiprLight {
    float data[11];
    uint32_t flags;
} HiprLight;

enum { HIPR_LIGHT_NONE = 0, HIPR_LIGHT_SPHERE = 1, HIPR_LIGHT_DIRECTIONAL = 2, HIPR_LIGHT_ENVIRONMENT = 3,
       HIPR_LIGHT_PRESAMPLED_ENVIRONMENT = 4, HIPR_LIGHT_SPOT = 5, HIPR_LIGHT_TYPE_MASK = 7 };

/* OR/Types.h:116-119, 16 bytes: float3 position + octahedral SNORM16 normal
 * (encode BF/Math/OctahedralNormal.h:53-83, decode OR/Types.h:62-69). Positions are in
 * OBJECT space, the normal as well. */
typedef struct HiprVertexGeometry {
    float position[3];
    int16_t oct_normal[2];
} HiprVertexGeometry;

/* OR/Types.h:46-52 */
enum { HIPR_MESH_NORMALS = 1, HIPR_MESH_TEXCOORDS = 2, HIPR_MESH_TINTS = 4, HIPR_MESH_EMISSIVE = 8 };

/* One per mesh model. Replaces the OptiX Transform -> GeometryGroup -> GeometryInstance chain
 * and its ModelState / mesh_flags variables (OR/Renderer.cpp:138-182, OR/Types.h:477-480).
 * Offsets index the pooled attribute arrays of HiprSceneDesc; an unused attribute's offset is
 * ignored (gated by mesh_flags exactly as ORS/TriangleAttributes.cu:50-83). */
typedef struct HiprInstance {
    float object_to_world[12];  /* row-major 3x4, to_matrix3x4(Transform) (BF/Math/Conversions.h:69-75) */
    uint32_t index_offset;      /* first uint3 of this mesh in `indices` */
    uint32_t vertex_offset;     /* first vertex of this mesh in `geometry` / `texcoords` / `tints` / `emissions` */
    int32_t instance_id;        /* InstanceID bits: (1 << 30) | mesh model index (OR/Types.h:121-138) */
    int32_t material_index;
    uint32_t mesh_flags;
    uint32_t _pad[3];
} HiprInstance;

/* World-space triangle in BVH leaf order, 48 bytes. Built by the host flattening the
 * instances (replaces the OptiX "Trbvh" build, OR/Renderer.cpp:161-182,471-476). */
typedef struct HiprTriangle {
    float v0[3];
    float v1[3];
    float v2[3];
    uint32_t instance_index;   /* index into HiprSceneDesc::instances */
    uint32_t primitive_index;  /* rtGetPrimitiveIndex() equivalent within the mesh */
    uint32_t flags;            /* HIPR_TRIANGLE_* */
} HiprTriangle;

enum {
    HIPR_TRIANGLE_OPAQUE = 1,    /* shadow rays terminate here: coverage is statically 1 (ORS/MonteCarlo.cu:278-285) */
    /* 2 is taken inside the device library (a flag of the exhaustive-search items) */
    HIPR_TRIANGLE_ONE_SIDED = 4  /* the hit program refuses a closest hit that arrives from behind and the path's ray is traced again from just past it
                                    (ORS/MonteCarlo.cu:147-164: the material is neither thin-walled nor a cut-out nor transmissive). The 8-wide search uses it to
                                    step over such a hit inside the traversal (hipr_set_backface_culling). May only be set where the instance's material is
                                    one-sided (validated); leaving it unset is always correct. */
};

/* BVH2 node, 64 bytes. Child c spans lo/hi boxes stored Aila-Laine style:
 *   c0xy = { c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y }
 *   c1xy = { c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y }
 *   cz   = { c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z }
 * child >= 0 : index of an inner node.
 * child <  0 : leaf, ~child = (first_triangle << 3) | (triangle_count - 1), count in [1, 8]. */
typedef struct HiprBvhNode {
    float c0xy[4];
    float c1xy[4];
    float cz[4];
    int32_t child[2];
    uint32_t _pad[2];
} HiprBvhNode;

/* Compressed 4-wide BVH node, 64 bytes: what the persistent traversal kernels walk in scenes with more than 64 BVH2 nodes.
 * (The traversal is bound by the number of 16-byte gathers it issues, not by arithmetic: one of these replaces ~2.1 BVH2
 * nodes of 64 bytes each.) Built by collapsing the BVH2 (host/BvhBuilder.cpp); child boxes are quantised to 8 bits per bound
 * relative to the node's own box, conservatively (decoded boxes contain the exact ones):
 *   lo_a[k] = origin[a] + qlo_a[k] * 2^(exponent_a - 127),  hi_a[k] = origin[a] + qhi_a[k] * 2^(exponent_a - 127)
 * exponents = ex | ey << 8 | ez << 16 (biased like an IEEE exponent field); byte k of qlo/qhi word a = child k.
 * child[k]: >= 0 inner wide node, < 0 leaf (same encoding as HiprBvhNode), HIPR_WIDE_EMPTY = unused slot. */
typedef struct HiprWideNode {
    float origin[3];
    uint32_t exponents;
    uint32_t qlo[3];
    uint32_t qhi[3];
    uint32_t _pad[2];
    int32_t child[4];
} HiprWideNode;
#define HIPR_WIDE_EMPTY 0x7FFFFFFF

/* 8-wide compressed BVH ("wide8"): what the persistent traversal kernels walk (round 3). Measured on the MI355X, the traversal is bound by the
 * number of 16-byte lane-loads the CU's address / tag pipeline takes (82 % busy with the 4-wide tree) and by the length of a ray's chain of
 * dependent fetches, so the tree is built to need few of both: ONE array of 64-byte SLOTS holds inner nodes and leaf records alike, the children of
 * a node sit in consecutive slots (one base index serves all eight), an inner node carries the 8-bit quantised boxes of up to EIGHT children in
 * four 16-byte loads, and a leaf is one record of one or two triangles sharing an edge in four loads (two triangles used to be six).
 * Node origin: three 21-bit coordinates on a grid over the scene's bounds, origin_a = fma(float(m_a), grid_cell[a], grid_min[a]); child box of
 * position s on axis a: [origin_a + qlo[a][s] * 2^(exponent[a] - 127), origin_a + qhi[a][s] * 2^(exponent[a] - 127)], containing the exact box.
 * Positions are assigned so that position bit a set means "on the + side of the node's centre along axis a": a ray visits the hit children in
 * ascending order of (position XOR the ray's octant bits, bit a = direction[a] < 0) -- near to far without sorting distances. */
typedef struct HiprNode8 {
    uint32_t origin[2];     /* x: bits 0..20, y: bits 21..41, z: bits 42..62 of the 64-bit value */
    uint8_t exponent[3];    /* biased like an IEEE exponent field */
    uint8_t inner_mask;     /* bit s: the child in position s is an inner node, else a leaf record */
    uint32_t base_valid;    /* bits 0..23: slot of the node's first child; bits 24..31: bit s = position s holds a child. The child in position s
                               lives in slot base + popcount(valid & ((1 << s) - 1)) */
    uint8_t qlo[3][8];      /* [axis][position]; an empty position holds qlo = 255, qhi = 0 */
    uint8_t qhi[3][8];
} HiprNode8;

/* Leaf record of the wide8 tree: triangle A = (a, a + e1, a + e2) and, when triangle[1] != HIPR_LEAF8_NONE, triangle B = (a, a + e2, a + e3): two
 * triangles of HiprSceneDesc::triangles that share the edge (a, a + e2). Both are tested with the ray / triangle solve of DESIGN.md on the stored
 * corner and edges (the edges rounded once, at build time), A first. The solve of a record triangle yields the weights (w, u, v) of its corners in
 * record order; `selectors` says which of them are the weights of the scene triangle's vertices 1 and 2, i.e. the (u, v) a hit reports. */
typedef struct HiprLeaf8 {
    float a[3], e1[3], e2[3], e3[3];
    uint32_t triangle[2];   /* indices into HiprSceneDesc::triangles */
    uint32_t flags;         /* bit 0 / bit 1: A / B is HIPR_TRIANGLE_OPAQUE; bit 2 / bit 3: A / B is HIPR_TRIANGLE_ONE_SIDED; bit 4: the scene triangle B is wound
                               against the record's (a, a + e2, a + e3) -- A never is --; bits 8..9, 10..11: which of (w, u, v) = 0, 1, 2 is A's reported u, v;
                               bits 12..13, 14..15: B's */
    float facing_margin;    /* a closest hit on a one-sided record triangle is stepped over when the solve's determinant, signed by the scene triangle's
                               winding, is below -facing_margin: the ray arrives from behind by more than any rounding of this test or of the hit program's
                               own (normalised geometric normal . direction < 0) could turn around. 2^-13 (|e1|^2 + |e2|^2 + |e3|^2) for unit directions. */
} HiprLeaf8;
#define HIPR_LEAF8_NONE 0xFFFFFFFFu
typedef union HiprSlot8 { HiprNode8 node; HiprLeaf8 leaf; uint32_t words[16]; } HiprSlot8;

/* Software replacement for the reference's texture samplers (OR/Renderer.cpp:703-751).
 * Texels live in HiprSceneDesc::texels at `texel_offset` (bytes). */
typedef struct HiprTexture {
    uint32_t width, height;
    uint64_t texel_offset; /* 64 bit: a Sponza-class set of 4K maps exceeds 4 GiB once expanded to RGBA */
    uint8_t format;        /* HIPR_TEXEL_* */
    uint8_t wrap_u, wrap_v;/* 0 clamp, 1 repeat (BF/Assets/Texture.h:21-35) */
    uint8_t filter;        /* bit0: linear magnification, bit1: linear minification */
    uint8_t is_sRGB;       /* RT_TEXTURE_READ_NORMALIZED_FLOAT_SRGB (OR/Renderer.cpp:736-739) */
    uint8_t _pad[3];
} HiprTexture;

enum { HIPR_TEXEL_R8 = 1, HIPR_TEXEL_RGBA8 = 4, HIPR_TEXEL_R32F = 17, HIPR_TEXEL_RGBA32F = 20 };

/* OR/Types.h:265-272 LightSample: what sampling a light returns. The environment light is PRE-sampled on the host. */
typedef struct HiprLightSample {
    float radiance[3];
    float PDF;
    float direction_to_light[3];
    float distance;
} HiprLightSample;

/* Importance sampled latitude-longitude environment map (PresampledEnvironmentLight, OR/Types.h:270-277, built by
 * OR/PresampledEnvironmentMap.cpp:19-101): the image is HiprSceneDesc::textures[environment_map_ID] (RGBA8 or RGBA32F);
 * `per_pixel_PDF` is the solid angle PDF of every PDF texel without its 1 / sin(theta) factor (nearest lookup, clamp);
 * `samples` are sample_count (a power of two >= 2) light samples drawn on the host from progressive multi-jittered points.
 * The environment takes part in next event estimation through a light of type HIPR_LIGHT_PRESAMPLED_ENVIRONMENT in `lights`
 * (the host appends it, OR/Renderer.cpp:1160-1196); its radiance is scaled by HiprSceneState::environment_tint. */
typedef struct HiprEnvironment {
    int32_t environment_map_ID;
    uint32_t pdf_width, pdf_height;
    const float* per_pixel_PDF;
    const HiprLightSample* samples;
    uint32_t sample_count;
} HiprEnvironment;

/* The flat scene the host produces in handle_updates() (OR/Renderer.cpp:578-1205). */
typedef struct HiprSceneDesc {
    const HiprBvhNode* nodes;            uint32_t node_count;
    const HiprTriangle* triangles;       uint32_t triangle_count;
    const HiprInstance* instances;       uint32_t instance_count;
    const uint32_t* indices;             uint32_t index_count;      /* uint3 per primitive, count in uints */
    const HiprVertexGeometry* geometry;  uint32_t vertex_count;
    const float* texcoords;              /* float2 per vertex or NULL */
    const uint32_t* tints;               /* uchar4 (tint rgb, roughness) per vertex or NULL */
    const float* emissions;              /* float3 per vertex or NULL */
    const HiprMaterial* materials;       uint32_t material_count;   /* slot 0 = invalid material */
    const HiprLight* lights;             uint32_t light_count;
    const HiprTexture* textures;         uint32_t texture_count;    /* slot 0 = none */
    const uint8_t* texels;               uint64_t texel_bytes;
    uint32_t bvh_max_depth;              /* deepest leaf, root = 1; selects the LDS stack size */
    const HiprWideNode* wide_nodes;      uint32_t wide_node_count;  /* the same tree collapsed to 4-wide nodes; may be NULL / 0 */
    uint32_t wide_stack_entries;         /* most entries a traversal of wide_nodes can have on its stack */
    const HiprEnvironment* environment;  /* NULL: the environment is the constant HiprSceneState::environment_tint */
    const HiprSlot8* wide8_slots;        uint32_t wide8_slot_count; /* the 8-wide tree over the same triangles, slot 0 = the root node; may be NULL / 0 */
    uint32_t wide8_height;               /* nodes on the longest root-to-leaf chain = most entries a traversal can have on its stack */
    float wide8_grid_min[3], wide8_grid_cell[3];
} HiprSceneDesc;

/* OR/Types.h:507-523 SceneStateGPU, without OptiX buffer ids. */
typedef struct HiprSceneState {
    float environment_tint[3];
    int32_t next_event_sample_count;   /* default 3 (OR/Renderer.cpp:479), clamped to 256 (:1390-1392) */
} HiprSceneState;

/* OR/Types.h:486-501 CameraStateGPU, matrices row-major as Matrix4x4f::begin(). */
typedef struct HiprCameraState {
    float view_to_world_rotation[9];
    float inverse_projection_matrix[16];
    float inverse_view_projection_matrix[16];
    uint32_t accumulations;
    uint32_t max_bounce_count;
    /* PathRegularizationSettings (OR/PublicTypes.h:38-45). The scale applied to the paths of accumulation a is
     * PDF_scale * (1.0f + scale_decay * float(a)), PublicTypes.h:44 PDF_scale_at_accumulation: the reference evaluates it on the
     * host once per launch (OR/Renderer.cpp:1244); a pass here may carry several accumulations, so the kernel evaluates it per path. */
    float path_regularization_PDF_scale;
    float path_regularization_scale_decay;
} HiprCameraState;

/* Precomputed tables uploaded at init (OR/Renderer.cpp:380-467). Float inputs are quantised to
 * unorm16 by the library exactly as the reference (`unsigned short(v * 65535 + 0.5f)`). */
typedef struct HiprTables {
    const float* ggx_with_fresnel_rho;  /* 32x32, F0 = 0, "base" */
    const float* ggx_rho;               /* 32x32, F0 = 1, "full" */
    const float* dielectric_light_rho;  /* 16x16x16 float2 (total, reflected), ior in [0.331492, 0.789474] */
    const float* dielectric_dense_rho;  /* 16x16x16 float2, ior in [1.26667, 3.01667] */
    const float* ggx_alpha_from_max_PDF;/* 32x32, x = encoded PDF, y = cos_theta */
} HiprTables;

/* Which pixels this context renders. The framebuffer is cut into 8x8 pixel tiles (one
 * wavefront of camera rays each), numbered row-major; the context owns tiles with
 * tile_id % tile_stride == tile_phase. tile_stride = 1 is the single GPU case. */
typedef struct HiprFrameDesc {
    uint32_t width, height;
    uint32_t tile_phase, tile_stride;
    uint32_t samples_per_pass;   /* accumulations traced per hipr_render_pass, >= 1 */
} HiprFrameDesc;

/* Ray / traversal counters accumulated since hipr_reset_counters(). */
typedef struct HiprCounters {
    uint64_t camera_rays;        /* P: pixel-samples generated */
    uint64_t closest_rays;       /* R_mc: closest-hit traces incl. retraces */
    uint64_t shadow_rays;        /* R_sh */
    uint64_t shaded_hits;        /* H: accepted surface hits */
    uint64_t closest_nodes;      /* N_mc (only when instrumented counting is enabled) */
    uint64_t closest_triangles;  /* T_mc */
    uint64_t shadow_nodes;       /* N_sh */
    uint64_t shadow_triangles;   /* T_sh */
    uint64_t iterations;         /* trace/shade rounds of the wavefront loop */
} HiprCounters;

/* Per-kernel device time of the passes since hipr_reset_timers(), measured with HIP events
 * on the context's stream. Index with HIPR_KERNEL_*. */
enum { HIPR_KERNEL_GENERATE = 0, HIPR_KERNEL_TRACE_CLOSEST = 1, HIPR_KERNEL_SHADE = 2,
       HIPR_KERNEL_TRACE_SHADOW = 3, HIPR_KERNEL_ACCUMULATE = 4, HIPR_KERNEL_COUNT = 5 };
typedef struct HiprKernelTimes {
    double milliseconds[HIPR_KERNEL_COUNT];
    uint64_t launches[HIPR_KERNEL_COUNT];
} HiprKernelTimes;

typedef struct HiprContext HiprContext;

/* ------------------------------------------------------------------------------------------- */
/* Lifetime: Renderer::initialize / ~Renderer (OR/Renderer.cpp:273-574, 1365-1387)             */
/* ------------------------------------------------------------------------------------------- */
int hipr_create(int device_id, HiprContext** out_context);
int hipr_destroy(HiprContext* context);
const char* hipr_last_error(void);
/* Number of HIP devices visible; 0 mirrors Context::getDeviceCount() == 0 (OR/Renderer.cpp:280). */
int hipr_device_count(void);

/* Use an externally owned hipStream_t (e.g. torch's current stream); NULL = the context's own. */
int hipr_set_stream(HiprContext* context, void* hip_stream);

/* ------------------------------------------------------------------------------------------- */
/* Uploads: the tables of OR/Renderer.cpp:380-467 and the scene of handle_updates :578-1205     */
/* ------------------------------------------------------------------------------------------- */
int hipr_upload_tables(HiprContext* context, const HiprTables* tables);
int hipr_upload_scene(HiprContext* context, const HiprSceneDesc* scene);
/* The host-side checks hipr_upload_scene runs before anything reaches the device, on their own (no context, no GPU): every index
 * the kernels follow -- material texture IDs, instance material / pool offsets, triangle instance / primitive / vertex indices,
 * texture extents inside the texel pool, BVH2 and wide BVH child and leaf ranges -- must be in range; HIPR_ERROR_INVALID_ARGUMENT
 * and a message in hipr_last_error() otherwise. (OptiX validates its node graph in rtContextValidate; this is the counterpart.) */
int hipr_validate_scene(const HiprSceneDesc* scene);
/* Transform-only update: the same scene after its host refitted the BVH to moved instances (the reference refits its root acceleration
 * structure when a node moves, OR/Renderer.cpp:472,1010-1041). Re-uploads nodes, wide nodes, triangles, instances and lights, whose
 * counts and tree topology must equal the uploaded scene's, and rebuilds the per-triangle records derived from them; meshes,
 * materials, textures, environment and tables stay as uploaded. */
int hipr_update_scene_geometry(HiprContext* context, const HiprSceneDesc* scene);
int hipr_set_scene_state(HiprContext* context, const HiprSceneState* state);

/* Entry points, numbered like OR/Types.h:33-44. set_backend() of the host renderer maps Backend values onto them
 * (OR/Renderer.cpp:1417-1455). The two AI denoiser entries (1, 2) wrap NVIDIA's proprietary DL denoiser and are not
 * provided: hipr_set_entry_point returns HIPR_ERROR_UNSUPPORTED for them. */
enum { HIPR_ENTRY_PATH_TRACING = 0, HIPR_ENTRY_DEPTH = 3, HIPR_ENTRY_ALBEDO = 4, HIPR_ENTRY_TINT = 5, HIPR_ENTRY_ROUGHNESS = 6,
       HIPR_ENTRY_SHADING_NORMAL = 7, HIPR_ENTRY_PRIMITIVE_ID = 8,
       /* The feature image the reference accumulates next to the radiance for its denoiser (AIDenoiser::path_tracing_RPG, ORS/SimpleRGPs.cu:149-201): at the
        * first accepted surface hit DefaultShading::rho of the material whatever its shading model, at a light hit radiance / (1 + radiance), black on a miss. */
       HIPR_ENTRY_DENOISER_ALBEDO = 9 };
int hipr_set_entry_point(HiprContext* context, int entry);
/* request_auxiliary_buffers (OR/Renderer.cpp:1267-1358) renders AOVs into scratch buffers so the camera's accumulation
 * survives: enable != 0 redirects hipr_render_pass / hipr_read_accumulation to a scratch accumulation buffer. enable == 1 zeroes it;
 * enable == 2 keeps what it holds (a second running mean next to the camera's: the denoiser's albedo feature image). */
int hipr_use_scratch_accumulation(HiprContext* context, int enable);

/* (Re)allocates the f64 accumulation buffer and the wavefront queues for the owned tiles;
 * resets nothing else. Replaces accumulation_buffer->setSize (OR/Renderer.cpp:1215-1219). */
int hipr_set_frame(HiprContext* context, const HiprFrameDesc* frame);
/* Number of pixels this context owns under the current frame description. */
int hipr_owned_pixel_count(HiprContext* context, uint32_t* out_count);

/* ------------------------------------------------------------------------------------------- */
/* Rendering: context->launch(entry, w, h) of OR/IBackend.h:37-39 / OR/Renderer.cpp:1250-1265   */
/* ------------------------------------------------------------------------------------------- */
/* Traces frame.samples_per_pass accumulations starting at camera->accumulations for every owned
 * pixel, folds them into the f64 running mean (ORS/SimpleRGPs.cu:74-107) and writes half4 pixels.
 * out_half4_device:
 *   tile_stride == 1 : full frame, pixel (x, y) at out[x + y * out_pitch_pixels]  (ORS/SimpleRGPs.cu:106)
 *   tile_stride  > 1 : compact, owned pixel k at out[k] in owned-tile-major order (see hipr_scatter_tiles)
 * May be NULL to skip the half4 output. Asynchronous on the context stream unless `synchronize`. */
int hipr_render_pass(HiprContext* context, const HiprCameraState* camera,
                     void* out_half4_device, uint32_t out_pitch_pixels, int synchronize);

/* hipr_render_pass in its two halves, for hosts that show a progressive image one accumulation at a time (the reference's
 * render() contract, OR/Renderer.cpp:1250-1265) without giving up batched tracing:
 *   hipr_trace_pass           traces frame.samples_per_pass accumulations starting at camera->accumulations and leaves the radiance
 *                             of every sample in the pass's per-sample buffer; the running mean is not touched;
 *   hipr_accumulate_samples   folds samples [first_sample, first_sample + sample_count) of the traced pass into the f64 running mean
 *                             as accumulations first_accumulation, first_accumulation + 1, ... and writes the half4 pixels.
 * Tracing 32 accumulations in one pass and folding them one per frame gives, frame by frame, the bits that 32 one-accumulation
 * passes give (ORS/SimpleRGPs.cu:74-107 is applied per sample either way), at the throughput of the batched pass.
 * hipr_set_samples_per_pass changes the batch size of the current frame without resetting the accumulation (queues only grow). */
int hipr_set_samples_per_pass(HiprContext* context, uint32_t samples_per_pass);
int hipr_trace_pass(HiprContext* context, const HiprCameraState* camera);
int hipr_accumulate_samples(HiprContext* context, uint32_t first_sample, uint32_t sample_count, uint32_t first_accumulation,
                            void* out_half4_device, uint32_t out_pitch_pixels, int synchronize);

/* Copies the f64 accumulation (double4 per owned pixel, compact owned-tile-major order or full
 * frame row-major when tile_stride == 1) to host memory. Blocking. */
int hipr_read_accumulation(HiprContext* context, double* out_rgba, uint64_t capacity_pixels);
/* Scatter `rank_count` compact half4 buffers (as gathered over RCCL, rank r at
 * compact_device + r * pixels_per_rank_stride) into a full frame. Runs on the context stream. */
int hipr_scatter_tiles(HiprContext* context, const void* compact_half4_device, uint64_t pixels_per_rank_stride,
                       uint32_t rank_count, uint32_t width, uint32_t height,
                       void* out_half4_device, uint32_t out_pitch_pixels);

/* ------------------------------------------------------------------------------------------- */
/* Device groups: one frame on several GPUs of a node, from one process.          */
/* The reference is single device (OR/Renderer.cpp:289-291; interop is disabled above one        */
/* device, DX11OptiXAdaptor/Adaptor.cpp:83-90), so these have no counterpart there: they are the  */
/* same calls as above, fanned out over one HiprContext per device, plus the one exchange the    */
/* partition needs. Member i owns the 8x8 tiles with tile % size == i (scene and tables          */
/* replicated, f64 accumulation kept per member: no collective per sample); a frame is assembled */
/* on member 0's device by gathering the members' compact half4 tiles (RCCL send / recv over      */
/* xGMI, or peer-to-peer copies when RCCL is absent or members share a device) and               */
/* hipr_scatter_tiles. The image is bit-identical to the single-device one. A group of one       */
/* device forwards every call to its context.                                                    */
/* ------------------------------------------------------------------------------------------- */
typedef struct HiprGroup HiprGroup;
int hipr_group_create(const int* device_ids, uint32_t count, HiprGroup** out_group);
int hipr_group_destroy(HiprGroup* group);
uint32_t hipr_group_size(HiprGroup* group);
HiprContext* hipr_group_context(HiprGroup* group, uint32_t member);
const char* hipr_group_gather_description(HiprGroup* group);   /* which transport the gather uses (and that it fell back, after an exchange that stalled) */
int hipr_group_upload_tables(HiprGroup* group, const HiprTables* tables);
int hipr_group_upload_scene(HiprGroup* group, const HiprSceneDesc* scene);
int hipr_group_update_scene_geometry(HiprGroup* group, const HiprSceneDesc* scene);
int hipr_group_set_scene_state(HiprGroup* group, const HiprSceneState* state);
int hipr_group_set_entry_point(HiprGroup* group, int entry);
int hipr_group_use_scratch_accumulation(HiprGroup* group, int enable);
int hipr_group_set_frame(HiprGroup* group, uint32_t width, uint32_t height, uint32_t samples_per_pass);
int hipr_group_set_samples_per_pass(HiprGroup* group, uint32_t samples_per_pass);
/* One host thread per member runs hipr_trace_pass; returns when every member has queued its last bounce. */
int hipr_group_trace_pass(HiprGroup* group, const HiprCameraState* camera);
/* Every member folds the samples into its running mean; with an output buffer (on member 0's device, full frame, row pitch in
 * pixels) the compact tiles are gathered and the frame is assembled there. */
int hipr_group_accumulate_samples(HiprGroup* group, uint32_t first_sample, uint32_t sample_count, uint32_t first_accumulation,
                                  void* out_half4_device, uint32_t out_pitch_pixels, int synchronize);
/* The full frame's f64 accumulation, row-major, assembled on the host from the members' tiles. */
int hipr_group_read_accumulation(HiprGroup* group, double* out_rgba, uint64_t capacity_pixels);
int hipr_group_get_counters(HiprGroup* group, HiprCounters* out);   /* sums over the members */

/* ------------------------------------------------------------------------------------------- */
/* Presentation side: what DX11OptiXAdaptor::Adaptor does around Renderer::render                */
/* (extensions/DX11OptiXAdapter/DX11OptiXAdaptor/Adaptor.cpp:141-247), so that the host library   */
/* above this ABI links no GPU runtime.                                                          */
/* ------------------------------------------------------------------------------------------- */
/* The render target / back buffer the adaptor owns (Adaptor.cpp:227-247 resize_render_target). */
int hipr_device_malloc(HiprContext* context, uint64_t bytes, void** out_device_pointer);
int hipr_device_free(HiprContext* context, void* device_pointer);   /* waits for the context stream first */
int hipr_device_memset(HiprContext* context, void* device_pointer, int byte_value, uint64_t bytes);   /* on the context stream: the clear of the back buffer before the cameras composite */
/* optix::Buffer::map() of the non-interop path (Adaptor.cpp:159-166). Blocking. */
int hipr_copy_to_host(HiprContext* context, void* host, const void* device_pointer, uint64_t bytes);
/* The adaptor's full-screen blit (Adaptor.cpp:96-100): backbuffer[x + y * backbuffer_pitch] =
 * pixels[x + (height - y - 1) * pitch] for the width x height viewport. Runs on the context stream. */
int hipr_present_flipped(HiprContext* context, const void* pixels_half4_device, uint32_t pitch_pixels, uint32_t width, uint32_t height,
                         void* backbuffer_half4_device, uint32_t backbuffer_pitch_pixels);

int hipr_synchronize(HiprContext* context);
int hipr_get_counters(HiprContext* context, HiprCounters* out);
int hipr_reset_counters(HiprContext* context);
/* Enables per-ray node / triangle visit counting in the trace kernels (slower, off by default). */
/* Number of independent wavefronts a pass is split into (0..4; at most one per 65536 path slots of the frame). Each share of the path
 * slots runs its bounces on its own stream, so one shades while another traces; results are bit-identical for any count. 0 (the default, or
 * HIPR_WAVEFRONTS) chooses by scene: two for scenes traced by the exhaustive / BVH2 kernels (+9 % ... +27 % on the Cornell box), one for the
 * persistent wide-BVH kernels, which fill the machine alone (measured: no gain, and per-kernel timers then time kernels that ran alone). */
int hipr_set_wavefront_count(HiprContext* context, int count);
/* The number of wavefronts the next pass runs as under the current frame, scene and limit: with the limit at 0 the library takes two for the persistent
 * kernels from 2^24 path slots per pass on (where one wavefront's launch drains, the other's blocks move in: profiles/r04_ab_wavefronts.txt), one below,
 * two for the small-scene kernels. 0 before hipr_set_frame. */
int hipr_get_wavefront_count(HiprContext* context, int* out_count);
/* How the uploaded scene is traced (chosen from its size; HIPR_TRACE_VARIANT overrides for experiments):
 *   HIPR_TRACE_BVH2             BVH2 kernels, one ray per lane (k_trace_closest / k_trace_shadow)
 *   HIPR_TRACE_WIDE_PERSISTENT  more than 64 BVH2 nodes: persistent kernels over the compressed 4-wide BVH; from bounce 1 on the
 *                               closest-hit rays of a bounce and the shadow rays of the previous one share one fused launch,
 *                               timed under HIPR_KERNEL_TRACE_CLOSEST
 *   HIPR_TRACE_EXHAUSTIVE       at most 64 triangles: every ray tests every triangle (k_trace_*_small)
 *   HIPR_TRACE_WIDE8_PERSISTENT more than 64 BVH2 nodes and HiprSceneDesc::wide8_slots given (the default then): persistent kernels over the 8-wide
 *                               tree with leaf records (k_trace_wide8), fused launches like HIPR_TRACE_WIDE_PERSISTENT
 * The CPU oracle has the same three searches; parity tests pick the matching one. */
enum { HIPR_TRACE_BVH2 = 0, HIPR_TRACE_WIDE_PERSISTENT = 1, HIPR_TRACE_EXHAUSTIVE = 2, HIPR_TRACE_WIDE8_PERSISTENT = 3 };
int hipr_get_trace_variant(HiprContext* context, int* out_variant);
/* Forces one of the searches above for the scenes uploaded AFTER the call (-1: by scene size, the default). Meant for tests and A/B measurements: every
 * search returns the same hits up to the rounding of its triangle solve; the oracle restates each of them. */
int hipr_set_trace_variant(HiprContext* context, int variant);
/* Pipelined passes (off by default; scenes traced by the persistent kernels only): consecutive hipr_trace_pass calls alternate between two sets of queues,
 * radiance buffers and streams, and a pass whose live paths have dwindled to under 1/64 of its slots queues all bounces that can still follow without waiting
 * for their sizes and returns, so that its tail -- launches that each take as long as one traversal -- drains while the next pass's full-size launches run.
 * Frames and counters are those of unpipelined passes bit for bit (tested). Measured on the MI355X it does not pay: the persistent kernels fill the CUs, the
 * tail's launches wait for their blocks to retire, and the blind bounces add empty launches (material scene, 32 bounces: 23.4 -> 24.1 ms per step). */
int hipr_set_pass_pipelining(HiprContext* context, int enable);
/* Back sides of one-sided surfaces (on by default; scenes traced by the 8-wide search). The reference's hit program refuses a closest hit that reaches a
 * one-sided surface from behind and traces the path's ray again from just past it (ORS/MonteCarlo.cu:147-164); with this on, the traversal steps over such
 * a hit itself -- when the triangle carries HIPR_TRIANGLE_ONE_SIDED and the hit is behind by more than HiprLeaf8::facing_margin -- and goes on to the hit the
 * retrace would have found. Same paths, same frames (tested against the retracing search); one BVH query less per refused hit: 18 % of the closest-hit
 * queries of the atrium. Hits inside the margin still go to the hit program. The one difference: a refused triangle that COINCIDES with another surface at the
 * same distance no longer hides it (the retrace starts past both). 0 restores the retrace for every refused hit. Applies to every closest-hit query of
 * the 8-wide search, the stage-level hipr_debug_trace_closest included. */
int hipr_set_backface_culling(HiprContext* context, int enable);
/* Arithmetic of the shade stage (K3: attribute interpolation, BSDFs, lights, next event estimation; the other kernels are correctly rounded always).
 *   HIPR_ARITHMETIC_FAST  (default) division, square root, sin, cos and pow by the hardware's approximations, contraction allowed, denormals flushed: what the
 *                         reference's shading PTX is built with (nvcc --use_fast_math, extensions/OptiXRenderer/CMakeLists.txt:82-83).
 *   HIPR_ARITHMETIC_EXACT IEEE division and square root, one rounding per operation, sin / cos / pow as the specified binary64 sequences of csrc/spec_math.h:
 *                         every pixel of every frame equals the CPU restatement (oracle/) in every bit, whatever the resolution or sample count -- the mode that
 *                         meets BASELINE.json's RMSE bound by construction (RMSE 0). About 10 % slower per step (bench.py config.exact_mode).
 * Both builds of the shade unit live in this library; the switch takes effect with the next pass (accumulate from zero after changing it: the two modes are
 * different, equally valid estimators of the same image). Environment HIPR_ARITHMETIC=exact|fast sets the default of new contexts.
 * Replaces a build-time choice of the reference (its CMake flag); no run-time counterpart there. */
enum { HIPR_ARITHMETIC_FAST = 0, HIPR_ARITHMETIC_EXACT = 1 };
int hipr_set_arithmetic(HiprContext* context, int arithmetic);
int hipr_get_arithmetic(HiprContext* context);      /* HIPR_ARITHMETIC_*, or a negative status */
int hipr_set_instrumentation(HiprContext* context, int count_traversal_steps);
int hipr_reset_timers(HiprContext* context);
int hipr_get_kernel_times(HiprContext* context, HiprKernelTimes* out);

/* ------------------------------------------------------------------------------------------- */
/* Stage-level entry points used by the parity tests: each runs ONE kernel of the path on       */
/* host-provided inputs and returns its raw output, so the kernels can be compared with the     */
/* CPU oracle bit for bit where the arithmetic is integer or correctly rounded f32.             */
/* ------------------------------------------------------------------------------------------- */
/* K1: camera rays for all owned pixels (ORS/SimpleRGPs.cu:44-72). Outputs host arrays of
 * owned_pixel_count entries: origin_tmin (float4), direction (float4, w unused), pixel (uint2 as x | y << 16). */
int hipr_debug_generate(HiprContext* context, const HiprCameraState* camera, uint32_t accumulation,
                        float* out_origin_tmin, float* out_direction, uint32_t* out_pixel);
/* RNG: PracticalScrambledSobol::sample4ui (OR/RNG.h:280-287) for n (accumulation, pixel_hash, dimension) triples. */
/* The shading models as the shade kernel evaluates them (DefaultShading / DiffuseShading / TransmissiveShading of
 * ORS/ShadingModels/), for the reference's golden vectors (ORT/ShadingModels/DefaultShadingTest.h:410-447,
 * TransmissiveShadingTest.h:203-236). shading_model: HIPR_SHADING_*. params10: tint[3], roughness, specularity, metallic,
 * coat, coat_roughness, cos_theta_o (NaN: wo.z), max_PDF_hint (NaN: none). mode 0: sample(wo, u = in) -> out7 = f[3], pdf,
 * direction[3]; mode 1: evaluate_with_PDF(wo, wi = in) -> out7 = f[3], pdf, 0, 0, 0. Host pointers. */
int hipr_debug_shading(HiprContext* context, int shading_model, const float* params10, const float* wo_n3, const float* in_n3, uint32_t n, int mode,
                       float* out_n7);
/* K3 at stage level: shade_path -- the closest-hit / miss / light-hit programs of ORS/MonteCarlo.cu:129-302 and ORS/SimpleRGPs.cu:349-362 as the shade kernel runs
 * them -- for n given queue entries of the uploaded scene: rays_n8 = origin, tmin, direction, BSDF PDF of the ray (raw: negative = delta dirac); throughput_bounces_n4 =
 * throughput, bits(bounces); hits_n4 = t, u, v, bits(triangle | 0x80000000 + light | 0xFFFFFFFF miss) as hipr_debug_trace_closest returns them; the triangle the ray
 * left from, the pixel's hash (pcg2d(x, y).x) and the accumulation of the sample. out_n32: per entry 0 flags (1 the path continues, 2 a shadow ray was emitted, 4 the hit
 * was shaded), 1-3 radiance added, 4-7 next origin + tmin, 8-11 next direction + BSDF PDF, 12-15 throughput + bits(bounces), 16 bits(last accepted triangle), 17-20
 * shadow origin + tmax, 21-23 direction to the light, 24-26 radiance the shadow ray carries; words of parts that do not apply are zero. Host pointers. The oracle's
 * hit programs write the same record (tests: bit-identical for the verification build, decision by decision statistics for the product). */
int hipr_debug_shade(HiprContext* context, const HiprCameraState* camera, uint32_t n, const float* rays_n8, const float* throughput_bounces_n4, const float* hits_n4,
                     const uint32_t* last_triangle, const uint32_t* pixel_hash, const uint32_t* accumulation, float* out_n32);
/* The light sources as the shade kernel evaluates them, for the reference's light tests (ORT/LightSources/SphereLightTest.h,
 * SpotLightTest.h). mode 0: LightSources::sample_radiance(light, position, u = in.xy) -> out8 = radiance[3], PDF,
 * direction_to_light[3], distance. mode 1 (spot lights only): out8 = evaluate(light, position, direction = in)[3],
 * pdf(light, position, direction), 0, 0, 0, 0. Host pointers; position3 is shared by the n inputs. */
int hipr_debug_light(HiprContext* context, const HiprLight* light, const float* position3, const float* in_n3, uint32_t n, int mode, float* out_n8);
/* The transcendentals of the shade unit in the context's arithmetic mode, over arrays: function 0 sin(x), 1 cos(x), 2 pow(x, y) (y ignored otherwise). In the exact
 * mode the results are csrc/spec_math.h's, bit for bit those of the oracle's restatement (tests/test_gpu_verify_build.py). */
int hipr_debug_math(HiprContext* context, int function, uint32_t n, const float* x, const float* y, float* out);
int hipr_debug_sobol(HiprContext* context, const uint32_t* accumulation_pixelhash_dimension, uint32_t n, uint32_t* out_uint4);
/* The VALU roof of the device the context runs on, measured (bench.py's roofline_valu; no reference counterpart): eight independent chains of one
 * instruction per lane, eight waves per SIMD on every CU, best of three launches. out3[0] = v_fma_f32, out3[1] = v_max_f32, out3[2] = v_cvt_f32_ubyte1,
 * each in wave64 instructions per second device-wide (tools/microbench/issue_rates.hip has the full table: profiles/r03_issue_rates.txt). */
int hipr_debug_valu_issue_rates(HiprContext* context, double* out3);
/* The 256 float4 reverse-Halton offsets the next event estimation candidates are drawn with, read back from the device
 * (g_random_sample_offsets, OR/Renderer.cpp:323-336). out_256x4: host pointer to 1024 floats. */
int hipr_debug_sample_offsets(HiprContext* context, float* out_256x4);
/* K2: closest hit for n rays. rays: float4 origin_tmin + float4 direction_tmax per ray; skip: global triangle index
 * to ignore per ray (0xFFFFFFFF = none). out_hits: float4 {t, u, v, bits(tri_or_light)} per ray. */
int hipr_debug_trace_closest(HiprContext* context, const float* rays, const uint32_t* skip, uint32_t n, float* out_hits);
/* K4: shadow transmittance for n rays (float4 origin_tmin + float4 direction_tmax) -> one float per ray. */
int hipr_debug_trace_shadow(HiprContext* context, const float* rays, uint32_t n, float* out_transmittance);

#ifdef __cplusplus
}
#endif

#endif /* HIPRENDERER_C_H */
