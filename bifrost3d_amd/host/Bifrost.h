// host/Bifrost.h -- the slice of the Bifrost core data model that a renderer plugin consumes.
//
// The reference core (core/Bifrost, 22 kLoC, MSVC-dialect C++) does not build outside Visual Studio
// (SURVEY.md 8c), so HIPRenderer is written against this mirror of exactly the surface
// OptiXRenderer::Renderer pulls from: static SoA managers with per-tick change
// sets, UID handles whose index 0 is the invalid sentinel, and the thin object wrappers. Names, argument
// meaning and the change-notification protocol follow the reference so host code reads the same:
//   BF = core/Bifrost/Bifrost
//   BF/Core/ChangeSet.h:22-78, BF/Core/UniqueIDGenerator.h:34-64
//   BF/Assets/{Mesh,Material,MeshModel,Image,Texture}.h, BF/Scene/{SceneNode,SceneRoot,Camera,LightSource}.h
// It is NOT a re-implementation of the engine: no serialization, no hierarchy beyond parent transforms,
// no image IO, single scene root semantics as used by the renderer.
#pragma once

#include "CameraEffects.h"
#include "Math.h"

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace Bifrost {

namespace Core {

// 24 bit index + 8 bit incarnation in the reference; the renderer only ever uses the index.
template <typename Tag>
struct UID {
    unsigned int id = 0;
    UID() = default;
    constexpr UID(unsigned int id) : id(id) {}
    static constexpr UID invalid_UID() { return UID(0u); }
    unsigned int get_index() const { return id; }
    operator unsigned int() const { return id; }
    bool operator==(UID rhs) const { return id == rhs.id; }
    bool operator!=(UID rhs) const { return id != rhs.id; }
};

// A set of change flags with the reference's query vocabulary (BF/Core/Bitmask.h).
template <typename E>
struct Bitmask {
    unsigned int mask = 0;
    Bitmask() = default;
    Bitmask(E e) : mask((unsigned int)e) {}
    Bitmask(std::initializer_list<E> es) { for (E e : es) mask |= (unsigned int)e; }
    unsigned int raw() const { return mask; }
    bool is_set(E e) const { return (mask & (unsigned int)e) != 0; }
    bool not_set(E e) const { return !is_set(e); }
    bool contains(E e) const { return (mask & (unsigned int)e) == (unsigned int)e; }
    template <typename... Es> bool any_set(Es... es) const { unsigned int m = 0; ((m |= (unsigned int)es), ...); return (mask & m) != 0; }
    bool operator==(E e) const { return mask == (unsigned int)e; }
    Bitmask& operator|=(E e) { mask |= (unsigned int)e; return *this; }
    bool is_empty() const { return mask == 0; }
};

// Change flags per ID plus the list of IDs changed since the last reset (BF/Core/ChangeSet.h).
template <typename ID, typename Change>
class ChangeSet {
public:
    void resize(size_t capacity) { m_changes.resize(capacity); }
    void add_change(ID id, Change change) {
        if (m_changes[id].is_empty()) m_changed.push_back(id);
        m_changes[id] |= change;
    }
    Bitmask<Change> get_changes(ID id) const { return id < m_changes.size() ? m_changes[id] : Bitmask<Change>(); }
    const std::vector<ID>& get_changed_resources() const { return m_changed; }
    void reset_change_notifications() {
        for (ID id : m_changed) m_changes[id] = Bitmask<Change>();
        m_changed.clear();
    }
private:
    std::vector<Bitmask<Change>> m_changes;
    std::vector<ID> m_changed;
};

// Iterable view with the is_empty() the reference's Iterable has.
template <typename ID>
struct Iterable {
    std::vector<ID> ids;
    typename std::vector<ID>::const_iterator begin() const { return ids.begin(); }
    typename std::vector<ID>::const_iterator end() const { return ids.end(); }
    bool is_empty() const { return ids.empty(); }
    size_t size() const { return ids.size(); }
};

// Storage shared by every manager: slot 0 is the invalid sentinel, destroyed slots are recycled.
template <typename ID, typename Record, typename Change>
class Manager {
public:
    Manager() { m_records.emplace_back(); m_alive.push_back(false); m_changes.resize(1); }
    unsigned int capacity() const { return (unsigned int)m_records.size(); }
    bool has(ID id) const { return id < m_alive.size() && m_alive[id]; }
    ID allocate() {
        ID id;
        if (!m_free.empty()) { id = m_free.back(); m_free.pop_back(); m_records[id] = Record(); }
        else { id = ID((unsigned int)m_records.size()); m_records.emplace_back(); m_alive.push_back(false); m_changes.resize(m_records.size()); }
        m_alive[id] = true;
        return id;
    }
    void release(ID id) { if (has(id)) { m_alive[id] = false; m_pending_free.push_back(id); } }
    Record& operator[](ID id) { return m_records[id]; }
    const Record& operator[](ID id) const { return m_records[id]; }
    Iterable<ID> get_iterable() const {
        Iterable<ID> it;
        for (unsigned int i = 1; i < m_alive.size(); ++i) if (m_alive[i]) it.ids.push_back(ID(i));
        return it;
    }
    void flag(ID id, Change c) { m_changes.add_change(id, c); }
    Bitmask<Change> get_changes(ID id) const { return m_changes.get_changes(id); }
    Iterable<ID> get_changed() const { return Iterable<ID>{m_changes.get_changed_resources()}; }
    void reset_change_notifications() {
        m_changes.reset_change_notifications();
        m_free.insert(m_free.end(), m_pending_free.begin(), m_pending_free.end());   // IDs are recycled after the tick, as in the reference
        m_pending_free.clear();
    }
    void clear() { *this = Manager(); }
private:
    std::vector<Record> m_records;
    std::vector<bool> m_alive;
    std::vector<ID> m_free, m_pending_free;
    ChangeSet<ID, Change> m_changes;
};

struct RenderersTag;
typedef UID<RenderersTag> RendererID;
class Renderers {
public:
    static RendererID create(std::string name) { names().push_back(name); return RendererID((unsigned int)names().size() - 1); }
    static void destroy(RendererID) {}
    static const std::string& get_name(RendererID id) { return names()[id]; }
private:
    static std::vector<std::string>& names() { static std::vector<std::string> n(1, "invalid"); return n; }
};

} // namespace Core

// =================================================================================================
// Assets
// =================================================================================================
namespace Assets {

using namespace Math;

enum class PixelFormat { Unknown = 0, Alpha8, Roughness8 = Alpha8, Intensity8, RGB24, RGBA32, Intensity_Float, RGB_Float, RGBA_Float };
inline int channel_count(PixelFormat f) {
    switch (f) { case PixelFormat::Alpha8: case PixelFormat::Intensity8: case PixelFormat::Intensity_Float: return 1;
                 case PixelFormat::RGB24: case PixelFormat::RGB_Float: return 3; case PixelFormat::RGBA32: case PixelFormat::RGBA_Float: return 4; default: return 0; }
}
enum class WrapMode { Clamp, Repeat };
enum class MagnificationFilter { None, Linear };
enum class MinificationFilter { None, Linear, Trilinear };

struct ImagesTag; typedef Core::UID<ImagesTag> ImageID;
class Images {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, PixelsUpdated = 4 };
    static ImageID create2D(std::string name, PixelFormat format, bool is_sRGB, unsigned width, unsigned height, const void* pixels, size_t byte_count) {
        ImageID id = m().allocate();
        Record& r = m()[id];
        r.name = name; r.format = format; r.is_sRGB = is_sRGB; r.width = width; r.height = height;
        r.pixels.assign((const uint8_t*)pixels, (const uint8_t*)pixels + byte_count);
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(ImageID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<ImageID> get_changed_images() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(ImageID id) { return m().get_changes(id); }
    static PixelFormat get_pixel_format(ImageID id) { return m()[id].format; }
    static bool is_sRGB(ImageID id) { return m()[id].is_sRGB; }
    static unsigned get_width(ImageID id) { return m()[id].width; }
    static unsigned get_height(ImageID id) { return m()[id].height; }
    static const void* get_pixels(ImageID id) { return m()[id].pixels.data(); }
    static void* get_mutable_pixels(ImageID id) { m().flag(id, Change::PixelsUpdated); return m()[id].pixels.data(); }
    static bool has(ImageID id) { return m().has(id); }
    static Core::Iterable<ImageID> get_iterable() { return m().get_iterable(); }
    static unsigned get_pixel_count(ImageID id) { return m()[id].width * m()[id].height; }
    static const std::string& get_name(ImageID id) { return m()[id].name; }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { std::string name; PixelFormat format = PixelFormat::Unknown; bool is_sRGB = false; unsigned width = 0, height = 0; std::vector<uint8_t> pixels; };
    static Core::Manager<ImageID, Record, Change>& m() { static Core::Manager<ImageID, Record, Change> s; return s; }
};

// Object wrapper (BF/Assets/Image.h), as much of it as the loaders use.
class Image {
public:
    Image(ImageID id = ImageID::invalid_UID()) : m_ID(id) {}
    static Image create2D(const std::string& name, PixelFormat format, bool is_sRGB, unsigned width, unsigned height, const void* pixels = nullptr) {
        const size_t bytes = size_t(width) * height * bytes_per_pixel(format);
        std::vector<uint8_t> zero;
        if (!pixels) { zero.assign(bytes, 0); pixels = zero.data(); }
        return Images::create2D(name, format, is_sRGB, width, height, pixels, bytes);
    }
    static size_t bytes_per_pixel(PixelFormat f) {
        switch (f) { case PixelFormat::Alpha8: case PixelFormat::Intensity8: return 1; case PixelFormat::RGB24: return 3; case PixelFormat::RGBA32: return 4;
                     case PixelFormat::Intensity_Float: return 4; case PixelFormat::RGB_Float: return 12; case PixelFormat::RGBA_Float: return 16; default: return 0; }
    }
    bool exists() const { return m_ID != ImageID::invalid_UID() && Images::has(m_ID); }
    ImageID get_ID() const { return m_ID; }
    PixelFormat get_pixel_format() const { return Images::get_pixel_format(m_ID); }
    unsigned get_width() const { return Images::get_width(m_ID); }
    unsigned get_height() const { return Images::get_height(m_ID); }
    unsigned get_pixel_count() const { return Images::get_pixel_count(m_ID); }
    const std::string& get_name() const { return Images::get_name(m_ID); }
    template <typename T> T* get_pixels() const { return static_cast<T*>(Images::get_mutable_pixels(m_ID)); }
private:
    ImageID m_ID;
};

struct TexturesTag; typedef Core::UID<TexturesTag> TextureID;
class Textures {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2 };
    static TextureID create2D(ImageID image, MagnificationFilter mag = MagnificationFilter::Linear, MinificationFilter min = MinificationFilter::Linear,
                              WrapMode u = WrapMode::Repeat, WrapMode v = WrapMode::Repeat) {
        TextureID id = m().allocate();
        m()[id] = {image, mag, min, u, v};
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(TextureID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<TextureID> get_changed_textures() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(TextureID id) { return m().get_changes(id); }
    static ImageID get_image_ID(TextureID id) { return m()[id].image; }
    static MagnificationFilter get_magnification_filter(TextureID id) { return m()[id].mag; }
    static MinificationFilter get_minification_filter(TextureID id) { return m()[id].min; }
    static WrapMode get_wrapmode_U(TextureID id) { return m()[id].u; }
    static WrapMode get_wrapmode_V(TextureID id) { return m()[id].v; }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { ImageID image; MagnificationFilter mag = MagnificationFilter::Linear; MinificationFilter min = MinificationFilter::Linear; WrapMode u = WrapMode::Repeat, v = WrapMode::Repeat; };
    static Core::Manager<TextureID, Record, Change>& m() { static Core::Manager<TextureID, Record, Change> s; return s; }
};

// ---- Materials (BF/Assets/Material.h) ----------------------------------------------------------------------------------
enum class MaterialFlag : unsigned char { None = 0, ThinWalled = 1, Cutout = 2 };
typedef Core::Bitmask<MaterialFlag> MaterialFlags;
enum class ShadingModel : unsigned char { Default = 0, Diffuse = 1, Transmissive = 2, Count = 3 };

const RGB iron_tint = RGB(0.560f, 0.570f, 0.580f);
const RGB gold_tint = RGB(1.000f, 0.766f, 0.336f);
const RGB copper_tint = RGB(0.955f, 0.637f, 0.538f);

struct MaterialsTag; typedef Core::UID<MaterialsTag> MaterialID;
class Materials {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, Updated = 4, ShadingModel = 8 };
    struct Data {
        MaterialFlags flags;
        ShadingModel shading_model = ShadingModel::Default;
        RGB tint = RGB(0.0f);
        TextureID tint_roughness_texture_ID;
        float roughness = 0, specularity = 0, metallic = 0;
        TextureID metallic_texture_ID;
        float coat = 0, coat_roughness = 0, coverage = 1;
        TextureID coverage_texture_ID;
        RGB emission = RGB(0.0f);
        static Data create_dielectric(RGB tint, float roughness, float specularity = 0.04f) { Data d; d.tint = tint; d.roughness = roughness; d.specularity = specularity; d.coverage = 1.0f; return d; }
        static Data create_metal(RGB tint, float roughness) { Data d; d.tint = tint; d.roughness = roughness; d.specularity = 1.0f; d.coverage = 1.0f; d.metallic = 1.0f; return d; }
        static Data create_coated_dielectric(RGB tint, float roughness, float specularity, float coat_roughness) {
            Data d = create_dielectric(tint, roughness, specularity); d.coat = 1.0f; d.coat_roughness = coat_roughness; return d;
        }
        static Data create_transmissive(RGB tint, float roughness, float specularity = 0.04f) { Data d = create_dielectric(tint, roughness, specularity); d.shading_model = ShadingModel::Transmissive; return d; }
    };
    static MaterialID create(std::string name, Data data) {
        MaterialID id = m().allocate();
        m()[id] = {name, data};
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(MaterialID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<MaterialID> get_iterable() { return m().get_iterable(); }
    static Core::Iterable<MaterialID> get_changed_materials() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(MaterialID id) { return m().get_changes(id); }
    static const Data& get_data(MaterialID id) { return m()[id].data; }
    static const std::string& get_name(MaterialID id) { return m()[id].name; }
    static void set_data(MaterialID id, const Data& d) {
        bool model_changed = d.shading_model != m()[id].data.shading_model;
        m()[id].data = d;
        m().flag(id, Change::Updated);
        if (model_changed) m().flag(id, Change::ShadingModel);
    }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { std::string name; Data data; };
    static Core::Manager<MaterialID, Record, Change>& m() { static Core::Manager<MaterialID, Record, Change> s; return s; }
};

// Object wrapper with the getters upload_material() reads (OptiXRenderer/Renderer.cpp:754-812).
class Material {
public:
    Material(MaterialID id = MaterialID::invalid_UID()) : m_ID(id) {}
    Material(const std::string& name, const Materials::Data& data) : m_ID(Materials::create(name, data)) {}
    static Material invalid() { return Material(); }
    bool exists() const { return m_ID != MaterialID::invalid_UID(); }
    const std::string& get_name() const { return Materials::get_name(m_ID); }
    static Material create_dielectric(const std::string& name, RGB tint, float roughness, float specularity = 0.04f) { return Materials::create(name, Materials::Data::create_dielectric(tint, roughness, specularity)); }
    static Material create_metal(const std::string& name, RGB tint, float roughness) { return Materials::create(name, Materials::Data::create_metal(tint, roughness)); }
    MaterialID get_ID() const { return m_ID; }
    MaterialFlags get_flags() const { return d().flags; }
    ShadingModel get_shading_model() const { return d().shading_model; }
    RGB get_tint() const { return d().tint; }
    float get_roughness() const { return d().roughness; }
    float get_specularity() const { return d().specularity; }
    float get_metallic() const { return d().metallic; }
    float get_coat() const { return d().coat; }
    float get_coat_roughness() const { return d().coat_roughness; }
    bool is_cutout() const { return d().flags.is_set(MaterialFlag::Cutout); }
    float get_coverage() const { return d().coverage; }
    float get_cutout_threshold() const { return d().coverage; }
    RGB get_emission() const { return d().emission; }
    TextureID get_tint_roughness_texture_ID() const { return d().tint_roughness_texture_ID; }
    TextureID get_metallic_texture_ID() const { return d().metallic_texture_ID; }
    TextureID get_coverage_texture_ID() const { return d().coverage_texture_ID; }
    // BF/Assets/Material.cpp:144-160: a tint texture has >= 3 channels, a roughness texture 4 channels or Roughness8.
    bool has_tint_texture() const { TextureID t = d().tint_roughness_texture_ID; return t != TextureID::invalid_UID() && channel_count(Images::get_pixel_format(Textures::get_image_ID(t))) >= 3; }
    bool has_roughness_texture() const {
        TextureID t = d().tint_roughness_texture_ID;
        if (t == TextureID::invalid_UID()) return false;
        PixelFormat f = Images::get_pixel_format(Textures::get_image_ID(t));
        return channel_count(f) == 4 || f == PixelFormat::Roughness8;
    }
    void set_flags(MaterialFlags f) { auto x = d(); x.flags = f; Materials::set_data(m_ID, x); }
    void set_shading_model(ShadingModel s) { auto x = d(); x.shading_model = s; Materials::set_data(m_ID, x); }
    void set_tint(RGB t) { auto x = d(); x.tint = t; Materials::set_data(m_ID, x); }
    void set_roughness(float r) { auto x = d(); x.roughness = r; Materials::set_data(m_ID, x); }
    Core::Bitmask<Materials::Change> get_changes() const { return Materials::get_changes(m_ID); }
    bool operator==(Material rhs) const { return m_ID == rhs.m_ID; }
private:
    const Materials::Data& d() const { return Materials::get_data(m_ID); }
    MaterialID m_ID;
};

// ---- Meshes (BF/Assets/Mesh.h) ----------------------------------------------------------------------------------------
enum class MeshFlag : unsigned char { None = 0, Position = 1, Normal = 2, Texcoord = 4, TintAndRoughness = 8, Emissive = 16,
                                      GeometryBuffers = 3, DefaultBuffers = 11, AllBuffers = 31 };
typedef Core::Bitmask<MeshFlag> MeshFlags;
struct TintRoughness { unsigned char r, g, b, roughness; };

struct MeshesTag; typedef Core::UID<MeshesTag> MeshID;
class Meshes {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2 };
    static MeshID create(std::string name, unsigned primitive_count, unsigned vertex_count, MeshFlags buffers = MeshFlag::AllBuffers) {
        MeshID id = m().allocate();
        Record& r = m()[id];
        r.name = name;
        r.primitives.resize(primitive_count);
        r.positions.resize(vertex_count);
        if (buffers.is_set(MeshFlag::Normal)) r.normals.resize(vertex_count);
        if (buffers.is_set(MeshFlag::Texcoord)) r.texcoords.resize(vertex_count);
        if (buffers.is_set(MeshFlag::TintAndRoughness)) r.tints.assign(vertex_count, TintRoughness{255, 255, 255, 255});
        if (buffers.is_set(MeshFlag::Emissive)) r.emission.resize(vertex_count);
        r.bounds = AABB::invalid();
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(MeshID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<MeshID> get_changed_meshes() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(MeshID id) { return m().get_changes(id); }
    static const std::string& get_name(MeshID id) { return m()[id].name; }
    static bool has(MeshID id) { return m().has(id); }
    static Core::Iterable<MeshID> get_iterable() { return m().get_iterable(); }
    static unsigned get_primitive_count(MeshID id) { return (unsigned)m()[id].primitives.size(); }
    static unsigned get_vertex_count(MeshID id) { return (unsigned)m()[id].positions.size(); }
    static Vector3ui* get_primitives(MeshID id) { return m()[id].primitives.data(); }
    static Vector3f* get_positions(MeshID id) { return m()[id].positions.data(); }
    static Vector3f* get_normals(MeshID id) { return m()[id].normals.empty() ? nullptr : m()[id].normals.data(); }
    static Vector2f* get_texcoords(MeshID id) { return m()[id].texcoords.empty() ? nullptr : m()[id].texcoords.data(); }
    static TintRoughness* get_tint_and_roughness(MeshID id) { return m()[id].tints.empty() ? nullptr : m()[id].tints.data(); }
    static Vector3f* get_emission(MeshID id) { return m()[id].emission.empty() ? nullptr : m()[id].emission.data(); }
    static AABB get_bounds(MeshID id) { return m()[id].bounds; }
    static void set_bounds(MeshID id, AABB b) { m()[id].bounds = b; }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { std::string name; std::vector<Vector3ui> primitives; std::vector<Vector3f> positions, normals, emission; std::vector<Vector2f> texcoords; std::vector<TintRoughness> tints; AABB bounds; };
    static Core::Manager<MeshID, Record, Change>& m() { static Core::Manager<MeshID, Record, Change> s; return s; }
};

class Mesh {
public:
    Mesh(MeshID id = MeshID::invalid_UID()) : m_ID(id) {}
    Mesh(const std::string& name, unsigned primitive_count, unsigned vertex_count, MeshFlags buffers = MeshFlag::AllBuffers) : m_ID(Meshes::create(name, primitive_count, vertex_count, buffers)) {}
    MeshID get_ID() const { return m_ID; }
    unsigned get_primitive_count() const { return Meshes::get_primitive_count(m_ID); }
    unsigned get_vertex_count() const { return Meshes::get_vertex_count(m_ID); }
    Vector3ui* get_primitives() const { return Meshes::get_primitives(m_ID); }
    Vector3f* get_positions() const { return Meshes::get_positions(m_ID); }
    Vector3f* get_normals() const { return Meshes::get_normals(m_ID); }
    Vector2f* get_texcoords() const { return Meshes::get_texcoords(m_ID); }
    TintRoughness* get_tint_and_roughness() const { return Meshes::get_tint_and_roughness(m_ID); }
    Vector3f* get_emission() const { return Meshes::get_emission(m_ID); }
    AABB get_bounds() const { return Meshes::get_bounds(m_ID); }
    void set_bounds(AABB b) { Meshes::set_bounds(m_ID, b); }
    const std::string& get_name() const { return Meshes::get_name(m_ID); }
    void compute_bounds() {   // BF/Assets/Mesh.cpp compute_bounds
        AABB b = AABB::invalid();
        for (unsigned v = 0; v < get_vertex_count(); ++v) b.grow_to_contain(get_positions()[v]);
        set_bounds(b);
    }
private:
    MeshID m_ID;
};

namespace MeshUtils {
Mesh deep_clone(Mesh mesh);                                     // BF/Assets/Mesh.cpp deep_clone: every buffer and the bounds copied
void transform_mesh(Mesh mesh, Matrix3x4f affine_transform);    // BF/Assets/Mesh.cpp:211-238: positions, bounds, normals by the inverse transpose
}

namespace MeshCreation {
Mesh plane(unsigned quads_per_edge, MeshFlags buffers = MeshFlag::AllBuffers);                              // BF/Assets/MeshCreation.cpp:30-75
Mesh box(unsigned quads_per_edge, Vector3f size = Vector3f::one(), MeshFlags buffers = MeshFlag::DefaultBuffers);   // :77-156
}

} // namespace Assets

// =================================================================================================
// Scene
// =================================================================================================
namespace Scene {

using namespace Math;

struct SceneNodesTag; typedef Core::UID<SceneNodesTag> SceneNodeID;
// Names are taken BY VALUE by every create(): the caller may pass a reference into the very storage that create() is about to grow
// (SceneNode(other.get_name(), ...): found by the address sanitizer run of the CPU suite).
class SceneNodes {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, Transform = 4 };
    static SceneNodeID create(std::string name, Transform transform = Transform::identity()) {
        SceneNodeID id = m().allocate();
        m()[id] = {name, transform, SceneNodeID::invalid_UID()};
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(SceneNodeID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<SceneNodeID> get_changed_nodes() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(SceneNodeID id) { return m().get_changes(id); }
    // Parenting keeps the GLOBAL transform of the child, like the reference (BF/Scene/SceneNode.cpp set_parent).
    static void set_parent(SceneNodeID id, SceneNodeID parent) { m()[id].parent = parent; }
    static SceneNodeID get_parent(SceneNodeID id) { return m()[id].parent; }
    static const std::string& get_name(SceneNodeID id) { return m()[id].name; }
    static bool has(SceneNodeID id) { return m().has(id); }
    static Transform get_global_transform(SceneNodeID id) { return m()[id].global_transform; }
    static void set_global_transform(SceneNodeID id, Transform t) {
        Transform delta = t * invert(m()[id].global_transform);
        m()[id].global_transform = t;
        m().flag(id, Change::Transform);
        for (SceneNodeID c : m().get_iterable())       // children follow
            if (m()[c].parent == id) set_global_transform(c, delta * m()[c].global_transform);
    }
    static std::vector<SceneNodeID> get_children(SceneNodeID id) { std::vector<SceneNodeID> r; for (SceneNodeID c : m().get_iterable()) if (m()[c].parent == id) r.push_back(c); return r; }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { std::string name; Transform global_transform = Transform::identity(); SceneNodeID parent; };
    static Core::Manager<SceneNodeID, Record, Change>& m() { static Core::Manager<SceneNodeID, Record, Change> s; return s; }
};

class SceneNode {
public:
    SceneNode(SceneNodeID id = SceneNodeID::invalid_UID()) : m_ID(id) {}
    SceneNode(const std::string& name, Transform t = Transform::identity()) : m_ID(SceneNodes::create(name, t)) {}
    static SceneNode invalid() { return SceneNode(); }
    bool operator==(SceneNode rhs) const { return m_ID == rhs.m_ID; }
    bool operator!=(SceneNode rhs) const { return !(m_ID == rhs.m_ID); }
    const std::string& get_name() const { return SceneNodes::get_name(m_ID); }
    SceneNode get_parent() const { return SceneNodes::get_parent(m_ID); }
    SceneNodeID get_ID() const { return m_ID; }
    void set_parent(SceneNode parent) { SceneNodes::set_parent(m_ID, parent.m_ID); }
    Transform get_global_transform() const { return SceneNodes::get_global_transform(m_ID); }
    void set_global_transform(Transform t) { SceneNodes::set_global_transform(m_ID, t); }
    std::vector<SceneNodeID> get_children() const { return SceneNodes::get_children(m_ID); }
private:
    SceneNodeID m_ID;
};

struct SceneRootsTag; typedef Core::UID<SceneRootsTag> SceneRootID;
class SceneRoots {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, EnvironmentTint = 4, EnvironmentMap = 8 };
    static SceneRootID create(std::string name, RGB environment_tint) {
        SceneRootID id = m().allocate();
        m()[id] = {name, environment_tint, SceneNodes::create(name + " root"), {}};
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(SceneRootID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static Core::Iterable<SceneRootID> get_changed_scenes() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(SceneRootID id) { return m().get_changes(id); }
    static RGB get_environment_tint(SceneRootID id) { return m()[id].tint; }
    static void set_environment_tint(SceneRootID id, RGB tint) { m()[id].tint = tint; m().flag(id, Change::EnvironmentTint); }
    // A latitude-longitude texture lighting the scene; the renderer importance samples it (BF/Scene/SceneRoot.h).
    static Assets::TextureID get_environment_map(SceneRootID id) { return m()[id].environment_map; }
    static void set_environment_map(SceneRootID id, Assets::TextureID map) { m()[id].environment_map = map; m().flag(id, Change::EnvironmentMap); }
    static Core::Iterable<SceneRootID> get_iterable() { return m().get_iterable(); }
    static SceneNodeID get_root_node(SceneRootID id) { return m()[id].root; }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { std::string name; RGB tint = RGB(0.0f); SceneNodeID root; Assets::TextureID environment_map; };
    static Core::Manager<SceneRootID, Record, Change>& m() { static Core::Manager<SceneRootID, Record, Change> s; return s; }
};

class SceneRoot {
public:
    SceneRoot(SceneRootID id = SceneRootID::invalid_UID()) : m_ID(id) {}
    SceneRoot(const std::string& name, RGB environment_tint) : m_ID(SceneRoots::create(name, environment_tint)) {}
    SceneRootID get_ID() const { return m_ID; }
    SceneNode get_root_node() const { return SceneRoots::get_root_node(m_ID); }
    RGB get_environment_tint() const { return SceneRoots::get_environment_tint(m_ID); }
    void set_environment_tint(RGB t) { SceneRoots::set_environment_tint(m_ID, t); }
    Assets::TextureID get_environment_map() const { return SceneRoots::get_environment_map(m_ID); }
    void set_environment_map(Assets::TextureID map) { SceneRoots::set_environment_map(m_ID, map); }
    Core::Bitmask<SceneRoots::Change> get_changes() const { return SceneRoots::get_changes(m_ID); }
private:
    SceneRootID m_ID;
};

struct CamerasTag; typedef Core::UID<CamerasTag> CameraID;
struct Screenshot {
    enum class Content : unsigned char { None = 0, ColorLDR = 1, ColorHDR = 2, Depth = 4, Albedo = 8, Tint = 16, Roughness = 32 };
    unsigned width = 0, height = 0;
    Content content = Content::None;
    Assets::PixelFormat format = Assets::PixelFormat::Unknown;
    void* pixels = nullptr;   // new[]-allocated, owned by the receiver (OptiXRenderer/Renderer.cpp:1308-1354)
};
class Cameras {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, Renderer = 4 };
    typedef Core::Bitmask<Screenshot::Content> ScreenshotContent;
    static CameraID create(std::string name, SceneRootID scene, Matrix4x4f projection, Matrix4x4f inverse_projection) {
        CameraID id = m().allocate();
        Record& r = m()[id];
        r.name = name; r.scene = scene; r.projection = projection; r.inverse_projection = inverse_projection; r.transform = Transform::identity();
        r.effects_settings = Math::CameraEffects::Settings::preset();     // Camera.cpp:157
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(CameraID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<CameraID> get_changed_cameras() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(CameraID id) { return m().get_changes(id); }
    static SceneRootID get_scene_ID(CameraID id) { return m()[id].scene; }
    static Core::RendererID get_renderer_ID(CameraID id) { return m()[id].renderer; }
    static void set_renderer_ID(CameraID id, Core::RendererID r) { m()[id].renderer = r; m().flag(id, Change::Renderer); }
    static Transform get_transform(CameraID id) { return m()[id].transform; }
    static void set_transform(CameraID id, Transform t) { m()[id].transform = t; }
    static Transform get_inverse_view_transform(CameraID id) { return m()[id].transform; }
    static Matrix4x4f get_projection_matrix(CameraID id) { return m()[id].projection; }
    static Matrix4x4f get_inverse_projection_matrix(CameraID id) { return m()[id].inverse_projection; }
    static void set_projection_matrices(CameraID id, Matrix4x4f p, Matrix4x4f ip) { m()[id].projection = p; m()[id].inverse_projection = ip; }
    static Matrix4x4f get_inverse_view_projection_matrix(CameraID id) { return to_matrix4x4(m()[id].transform) * m()[id].inverse_projection; }   // Camera.h:112-114
    // The part of the window the camera renders to, normalised (Camera.h:116-124); the whole window by default.
    static int get_z_index(CameraID id) { return m()[id].z_index; }
    static void set_z_index(CameraID id, int z_index) { m()[id].z_index = z_index; }
    static std::vector<CameraID> get_z_sorted_IDs() {      // BF/Scene/Camera.cpp:171-178
        std::vector<CameraID> IDs;
        for (CameraID id : get_iterable()) IDs.push_back(id);
        std::stable_sort(IDs.begin(), IDs.end(), [](CameraID lhs, CameraID rhs) { return get_z_index(lhs) < get_z_index(rhs); });
        return IDs;
    }
    static void set_viewport(CameraID id, float x, float y, float width, float height) { Record& r = m()[id]; r.viewport[0] = x; r.viewport[1] = y; r.viewport[2] = width; r.viewport[3] = height; }
    static void get_window_viewport(CameraID id, Vector2i window_size, int& x, int& y, int& width, int& height) {
        const Record& r = m()[id];
        x = int(r.viewport[0] * window_size.x); y = int(r.viewport[1] * window_size.y); width = int(r.viewport[2] * window_size.x); height = int(r.viewport[3] * window_size.y);
    }
    static Math::CameraEffects::Settings get_effects_settings(CameraID id) { return m()[id].effects_settings; }                 // Camera.h:126-127
    static void set_effects_settings(CameraID id, Math::CameraEffects::Settings settings) { m()[id].effects_settings = settings; }
    static Core::Iterable<CameraID> get_iterable() { return m().get_iterable(); }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record {
        std::string name; SceneRootID scene; Core::RendererID renderer; Transform transform = Transform::identity();
        Matrix4x4f projection = Matrix4x4f::identity(), inverse_projection = Matrix4x4f::identity();
        float viewport[4] = {0.0f, 0.0f, 1.0f, 1.0f};
        int z_index = 0;                                                   // Camera.h: cameras composite in ascending z-index
        Math::CameraEffects::Settings effects_settings = Math::CameraEffects::Settings::preset();
    };
    static Core::Manager<CameraID, Record, Change>& m() { static Core::Manager<CameraID, Record, Change> s; return s; }
};

namespace CameraUtils {
void compute_perspective_projection(float near_distance, float far_distance, float field_of_view_in_radians, float aspect_ratio, Matrix4x4f& projection, Matrix4x4f& inverse_projection);
void compute_orthographic_projection(float width, float height, float depth, Matrix4x4f& projection, Matrix4x4f& inverse_projection);
}

struct LightSourcesTag; typedef Core::UID<LightSourcesTag> LightSourceID;
class LightSources {
public:
    enum class Type : unsigned char { Sphere, Spot, Directional };
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, Updated = 4 };
    static LightSourceID create_sphere_light(SceneNodeID node, RGB power, float radius) { return make({Type::Sphere, node, power, radius, 1.0f}); }
    static LightSourceID create_spot_light(SceneNodeID node, RGB power, float radius, float cos_angle) { return make({Type::Spot, node, power, radius, cos_angle}); }
    static LightSourceID create_directional_light(SceneNodeID node, RGB radiance) { return make({Type::Directional, node, radiance, 0.0f, 1.0f}); }
    static void destroy(LightSourceID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<LightSourceID> get_iterable() { return m().get_iterable(); }
    static Core::Iterable<LightSourceID> get_changed_lights() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(LightSourceID id) { return m().get_changes(id); }
    static Type get_type(LightSourceID id) { return m()[id].type; }
    static SceneNodeID get_node_ID(LightSourceID id) { return m()[id].node; }
    static RGB get_power(LightSourceID id) { return m()[id].color; }       // radiance for directional lights
    static float get_radius(LightSourceID id) { return m()[id].radius; }
    static float get_cos_angle(LightSourceID id) { return m()[id].cos_angle; }
    static bool is_delta_light(LightSourceID id) { return m()[id].type == Type::Directional || m()[id].radius == 0.0f; }
    static void set_power(LightSourceID id, RGB p) { m()[id].color = p; m().flag(id, Change::Updated); }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { Type type = Type::Sphere; SceneNodeID node; RGB color = RGB(0.0f); float radius = 0, cos_angle = 1; };
    static LightSourceID make(Record r) { LightSourceID id = m().allocate(); m()[id] = r; m().flag(id, Change::Created); return id; }
    static Core::Manager<LightSourceID, Record, Change>& m() { static Core::Manager<LightSourceID, Record, Change> s; return s; }
};

// Convenience constructors with the reference's spelling: SphereLight(node, power, radius) etc.
inline LightSourceID SphereLight(SceneNode node, RGB power, float radius) { return LightSources::create_sphere_light(node.get_ID(), power, radius); }
inline LightSourceID SpotLight(SceneNode node, RGB power, float radius, float cos_angle) { return LightSources::create_spot_light(node.get_ID(), power, radius, cos_angle); }
inline LightSourceID DirectionalLight(SceneNode node, RGB radiance) { return LightSources::create_directional_light(node.get_ID(), radiance); }

} // namespace Scene

namespace Assets {

struct MeshModelsTag; typedef Core::UID<MeshModelsTag> MeshModelID;
class MeshModels {
public:
    enum class Change : unsigned char { None = 0, Created = 1, Destroyed = 2, Material = 4 };
    static MeshModelID create(Scene::SceneNodeID node, MeshID mesh, MaterialID material) {
        MeshModelID id = m().allocate();
        m()[id] = {node, mesh, material};
        m().flag(id, Change::Created);
        return id;
    }
    static void destroy(MeshModelID id) { if (m().has(id)) { m().flag(id, Change::Destroyed); m().release(id); } }
    static unsigned capacity() { return m().capacity(); }
    static Core::Iterable<MeshModelID> get_iterable() { return m().get_iterable(); }
    static Core::Iterable<MeshModelID> get_changed_models() { return m().get_changed(); }
    static Core::Bitmask<Change> get_changes(MeshModelID id) { return m().get_changes(id); }
    static Scene::SceneNodeID get_scene_node_ID(MeshModelID id) { return m()[id].node; }
    static MeshID get_mesh_ID(MeshModelID id) { return m()[id].mesh; }
    static MaterialID get_material_ID(MeshModelID id) { return m()[id].material; }
    static void set_material_ID(MeshModelID id, MaterialID material) { m()[id].material = material; m().flag(id, Change::Material); }
    static void reset_change_notifications() { m().reset_change_notifications(); }
    static void deallocate() { m().clear(); }
private:
    struct Record { Scene::SceneNodeID node; MeshID mesh; MaterialID material; };
    static Core::Manager<MeshModelID, Record, Change>& m() { static Core::Manager<MeshModelID, Record, Change> s; return s; }
};

class MeshModel {
public:
    MeshModel(MeshModelID id = MeshModelID::invalid_UID()) : m_ID(id) {}
    MeshModel(Scene::SceneNode node, Mesh mesh, Material material) : m_ID(MeshModels::create(node.get_ID(), mesh.get_ID(), material.get_ID())) {}
    MeshModelID get_ID() const { return m_ID; }
    Mesh get_mesh() const { return MeshModels::get_mesh_ID(m_ID); }
    Material get_material() const { return MeshModels::get_material_ID(m_ID); }
    Scene::SceneNode get_scene_node() const { return MeshModels::get_scene_node_ID(m_ID); }
    Core::Bitmask<MeshModels::Change> get_changes() const { return MeshModels::get_changes(m_ID); }
private:
    MeshModelID m_ID;
};

} // namespace Assets

// What the application's cleanup callback does after rendering a tick (apps/SimpleViewer/main.cpp:298-308).
inline void reset_all_change_notifications() {
    Assets::Images::reset_change_notifications();
    Assets::Textures::reset_change_notifications();
    Assets::Materials::reset_change_notifications();
    Assets::Meshes::reset_change_notifications();
    Assets::MeshModels::reset_change_notifications();
    Scene::SceneNodes::reset_change_notifications();
    Scene::SceneRoots::reset_change_notifications();
    Scene::Cameras::reset_change_notifications();
    Scene::LightSources::reset_change_notifications();
}

inline void deallocate_all() {
    Assets::Images::deallocate(); Assets::Textures::deallocate(); Assets::Materials::deallocate(); Assets::Meshes::deallocate();
    Assets::MeshModels::deallocate(); Scene::SceneNodes::deallocate(); Scene::SceneRoots::deallocate(); Scene::Cameras::deallocate();
    Scene::LightSources::deallocate();
}

} // namespace Bifrost
