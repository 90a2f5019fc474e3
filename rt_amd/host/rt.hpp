// rt.hpp — C++ host mirror of the reference's scene/camera API above the C ABI (include/rt_hip.h).
//
// The reference is compiled (Rust) code whose toolchain is absent from the build image, so compiled
// callers get this header-only mirror with the reference's names and argument meaning:
//
//   rt::Camera::create(...)            Camera::new            src/camera.rs:169-227
//   rt::Camera::render_image(world)    Camera::render_image   src/camera.rs:327-341
//   rt::World::build(shapes)           World::build           src/hittable.rs:33-46
//   rt::World::hit(rays, lo, hi)       World::hit             src/hittable.rs:135-149
//   rt::Sphere / rt::Triangle          src/hittable.rs:153-297
//   rt::Lambertian / Metal / Dielectric, SolidColor / CheckerTexture / ImageTexture
//                                      src/material.rs, src/texture.rs
//
// Shared materials (`Arc<Material>`) are std::shared_ptr and are de-duplicated by pointer identity
// when flattening, like the Rust shim would.  All arithmetic here is f64 in the reference's
// operation order (nalgebra: dot = (x0*y0 + x1*y1) + x2*y2, normalize = v / |v|); compile host code
// with -ffp-contract=off.  Errors are rt::Error exceptions on the C++ side only; nothing throws
// across the C ABI.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "../../include/rt_hip.h"

namespace rt {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != RT_OK) throw Error(rc, rt_last_error());
}

struct Vec3 {
    double x = 0, y = 0, z = 0;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec3 normalize(Vec3 a) { return a / std::sqrt(dot(a, a)); }
inline void store(Vec3 v, double* p) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

// ---- textures (src/texture.rs) ----------------------------------------------------------------
struct Texture;
using TexturePtr = std::shared_ptr<const Texture>;
struct Texture {
    uint32_t kind = RT_TEX_SOLID;
    Vec3 color;                       // SolidColor
    double scale_inverted = 1.0;      // CheckerTexture (1.0 / scale, src/texture.rs:54)
    TexturePtr even, odd;
    uint32_t width = 0, height = 0;   // ImageTexture: RGBA8 texels, row-major
    std::vector<uint8_t> rgba8;
};
inline TexturePtr SolidColor(Vec3 c) {
    auto t = std::make_shared<Texture>();
    t->kind = RT_TEX_SOLID; t->color = c;
    return t;
}
inline TexturePtr CheckerTexture(double scale, TexturePtr even, TexturePtr odd) {
    auto t = std::make_shared<Texture>();
    t->kind = RT_TEX_CHECKER; t->scale_inverted = 1.0 / scale; t->even = std::move(even); t->odd = std::move(odd);
    return t;
}
inline TexturePtr ImageTexture(uint32_t w, uint32_t h, std::vector<uint8_t> rgba8) {
    if (!w || !h || rgba8.size() != size_t(w) * h * 4) throw Error(RT_ERR_INVALID, "ImageTexture: bad size");
    auto t = std::make_shared<Texture>();
    t->kind = RT_TEX_IMAGE; t->width = w; t->height = h; t->rgba8 = std::move(rgba8);
    return t;
}

// ---- materials (src/material.rs) --------------------------------------------------------------
struct Material {
    uint32_t kind = RT_MAT_LAMBERTIAN;
    TexturePtr texture;
    bool has_fuzz = false;            // Option<Float>::is_some()
    double fuzz = 0.0, refractive_index = 1.0;
};
using MaterialPtr = std::shared_ptr<const Material>;
inline MaterialPtr Lambertian(TexturePtr t) {
    auto m = std::make_shared<Material>();
    m->kind = RT_MAT_LAMBERTIAN; m->texture = std::move(t);
    return m;
}
inline MaterialPtr Metal(TexturePtr t, const double* fuzz = nullptr) {
    auto m = std::make_shared<Material>();
    m->kind = RT_MAT_METAL; m->texture = std::move(t); m->has_fuzz = fuzz != nullptr; m->fuzz = fuzz ? *fuzz : 0.0;
    return m;
}
inline MaterialPtr Dielectric(double refractive_index, const double* fuzz = nullptr) {
    auto m = std::make_shared<Material>();
    m->kind = RT_MAT_DIELECTRIC; m->refractive_index = refractive_index; m->has_fuzz = fuzz != nullptr; m->fuzz = fuzz ? *fuzz : 0.0;
    return m;
}

// ---- shapes (src/hittable.rs) -------------------------------------------------------------------
struct Sphere {
    Vec3 center; double radius; MaterialPtr material; Vec3 front_direction{1.0, 0.0, 0.0};
    Sphere(Vec3 c, double r, MaterialPtr m) : center(c), radius(r > 0.0 ? r : 0.0), material(std::move(m)) {}
    static Sphere new_facing(Vec3 c, double r, MaterialPtr m, Vec3 front) {
        Sphere s(c, r, std::move(m)); s.front_direction = front; return s;
    }
};
struct Triangle {
    Vec3 a, b, c; double uv[3][2] = {{0.0, 0.0}, {1.0, 0.0}, {0.5, 1.0}}; Vec3 normal; MaterialPtr material;
    Triangle(Vec3 a_, Vec3 b_, Vec3 c_, MaterialPtr m) : a(a_), b(b_), c(c_), material(std::move(m)) {
        normal = normalize(cross(normalize(b - a), normalize(c - a)));       // src/hittable.rs:169-178
    }
    static Triangle new_opposite_normal(Vec3 a, Vec3 b, Vec3 c, MaterialPtr m) { return Triangle(c, b, a, std::move(m)); }
};
using Shape = std::variant<Sphere, Triangle>;

// scenes::generate_ground_plane (src/scenes.rs:310-342)
inline std::vector<Shape> generate_ground_plane(double width, double length, double z, MaterialPtr m, bool top_is_up) {
    double hw = width / 2.0, hl = length / 2.0;
    Vec3 a{-hw, -hl, z}, b{hw, -hl, z}, c{hw, hl, z}, d{-hw, hl, z};
    if (top_is_up) return {Triangle(a, b, c, m), Triangle(a, c, d, m)};
    return {Triangle::new_opposite_normal(a, b, c, m), Triangle::new_opposite_normal(a, c, d, m)};
}

// ---- World (src/hittable.rs:24-46, :135-149) ----------------------------------------------------
class World {
public:
    World(const World&) = delete;
    World& operator=(const World&) = delete;
    World(World&& o) noexcept : h_(o.h_), device_(o.device_), prim_shape_(std::move(o.prim_shape_)) { o.h_ = nullptr; }
    ~World() { rt_scene_destroy(h_); }

    // Flattens `shapes` ([spheres..., triangles...] keeps each kind's order) and uploads to `device`.
    static World build(const std::vector<Shape>& shapes, const rt_sky& sky, int device = 0) {
        std::vector<rt_sphere> S; std::vector<rt_triangle> T; std::vector<rt_material> M; std::vector<rt_texture> X;
        std::vector<uint8_t> pool;
        std::map<const Material*, uint32_t> mid; std::map<const Texture*, uint32_t> tid;
        auto tex = [&](auto&& self, const TexturePtr& t) -> uint32_t {
            if (!t) throw Error(RT_ERR_INVALID, "material without texture");
            auto it = tid.find(t.get());
            if (it != tid.end()) return it->second;
            rt_texture r; std::memset(&r, 0, sizeof(r));
            r.kind = t->kind;
            if (t->kind == RT_TEX_SOLID) store(t->color, r.rgb);
            else if (t->kind == RT_TEX_CHECKER) { r.even = self(self, t->even); r.odd = self(self, t->odd); r.inv_scale = t->scale_inverted; }
            else { r.width = t->width; r.height = t->height; r.texel_format = RT_TEXEL_RGBA8; r.texel_offset = pool.size();
                   pool.insert(pool.end(), t->rgba8.begin(), t->rgba8.end()); }
            X.push_back(r);
            return tid[t.get()] = uint32_t(X.size() - 1);
        };
        auto mat = [&](const MaterialPtr& m) -> uint32_t {
            if (!m) throw Error(RT_ERR_INVALID, "shape without material");
            auto it = mid.find(m.get());
            if (it != mid.end()) return it->second;
            rt_material r; std::memset(&r, 0, sizeof(r));
            r.kind = m->kind; r.has_fuzz = m->has_fuzz; r.fuzz = m->fuzz; r.ior = m->refractive_index;
            if (m->kind != RT_MAT_DIELECTRIC) r.texture = tex(tex, m->texture);
            M.push_back(r);
            return mid[m.get()] = uint32_t(M.size() - 1);
        };
        std::vector<uint32_t> sph_pos, tri_pos;        // where each primitive sat in `shapes`
        for (size_t si = 0; si < shapes.size(); ++si) {
            const Shape& s = shapes[si];
            (std::holds_alternative<Sphere>(s) ? sph_pos : tri_pos).push_back(uint32_t(si));
            if (const Sphere* sp = std::get_if<Sphere>(&s)) {
                rt_sphere r; std::memset(&r, 0, sizeof(r));
                store(sp->center, r.center); r.radius = sp->radius; store(sp->front_direction, r.face); r.material = mat(sp->material);
                S.push_back(r);
            } else {
                const Triangle& t = std::get<Triangle>(s);
                rt_triangle r; std::memset(&r, 0, sizeof(r));
                store(t.a, r.a); store(t.b, r.b); store(t.c, r.c); store(t.normal, r.n);
                std::memcpy(r.uv, t.uv, sizeof(r.uv)); r.material = mat(t.material);
                T.push_back(r);
            }
        }
        rt_scene* h = nullptr;
        check(rt_scene_create(S.data(), S.size(), T.data(), T.size(), M.data(), M.size(), X.data(), X.size(),
                              pool.data(), pool.size(), &sky, device, &h));
        World w(h, device);
        w.prim_shape_ = std::move(sph_pos);
        w.prim_shape_.insert(w.prim_shape_.end(), tri_pos.begin(), tri_pos.end());
        return w;
    }
    // Position in the `shapes` vector handed to build() of primitive id `prim` (the ABI orders primitives
    // [spheres..., triangles...]; this maps a hit back to the reference's Vec<Shape> index for mixed lists).
    uint32_t shape_of(uint32_t prim) const { return prim_shape_.at(prim); }

    std::vector<rt_hit> hit(const std::vector<rt_ray>& rays, double t_min = 0.001, double t_max = 1.7976931348623157e308) const {
        std::vector<rt_hit> out(rays.size());
        check(rt_trace_closest(h_, rays.data(), rays.size(), t_min, t_max, out.data(), nullptr));
        return out;
    }
    const rt_scene* handle() const { return h_; }
    int device() const { return device_; }

private:
    World(rt_scene* h, int device) : h_(h), device_(device) {}
    rt_scene* h_;
    int device_;
    std::vector<uint32_t> prim_shape_;
};

// ---- Camera (src/camera.rs) -----------------------------------------------------------------------
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<double> rgb;          // (y*width + x)*3 + c, linear mean radiance
};

class Camera {
public:
    rt_camera rec;
    uint32_t samples_per_pixel = 1;

    // Camera::new (src/camera.rs:169-227): same positional arguments, same operation order.
    static Camera create(Vec3 center, Vec3 lookat, Vec3 up, double focus_distance, double defocus_angle,
                         uint32_t image_width, uint32_t image_height, uint32_t samples_per_pixel, uint32_t max_depth,
                         double vertical_fov, double t_max = 1.7976931348623157e308) {
        const double rad = 3.14159265358979323846264338327950288 / 180.0;      // f64::to_radians
        Vec3 w = normalize(center - lookat), u = normalize(cross(up, w)), v = cross(w, u);
        double h = std::tan((vertical_fov * rad) / 2.0);
        double viewport_height = 2.0 * h * focus_distance;
        double aspect_ratio = double(image_width) / double(image_height);
        double viewport_width = viewport_height * aspect_ratio;
        Vec3 viewport_u = u * viewport_width, viewport_v = (-v) * viewport_height;
        Vec3 pixel_du = viewport_u / double(image_width), pixel_dv = viewport_v / double(image_height);
        Vec3 upper_left = center - (w * focus_distance) - viewport_u / 2.0 - viewport_v / 2.0;
        Vec3 pixel00 = upper_left + (pixel_du + pixel_dv) / 2.0;
        double defocus_radius = focus_distance * std::tan((defocus_angle / 2.0) * rad);
        Camera c; std::memset(&c.rec, 0, sizeof(c.rec));
        store(center, c.rec.center); store(pixel00, c.rec.pixel00); store(pixel_du, c.rec.pixel_du); store(pixel_dv, c.rec.pixel_dv);
        store(u * defocus_radius, c.rec.defocus_u); store(v * defocus_radius, c.rec.defocus_v);
        c.rec.defocus_angle = defocus_angle; c.rec.t_min = 0.001; c.rec.t_max = t_max;
        c.rec.width = image_width; c.rec.height = image_height; c.rec.max_depth = max_depth;
        c.samples_per_pixel = samples_per_pixel;
        return c;
    }

    // Camera::render_image (src/camera.rs:327-341)
    Image render_image(const World& world, uint64_t seed = 0, rt_stats* stats = nullptr) const {
        rt_opts o; std::memset(&o, 0, sizeof(o));
        o.spp = samples_per_pixel; o.sample_stride = 1; o.precision = RT_PRECISION_F64; o.seed = seed; o.device = world.device();
        Image img; img.width = rec.width; img.height = rec.height; img.rgb.resize(size_t(rec.width) * rec.height * 3);
        check(rt_render(world.handle(), &rec, &o, img.rgb.data(), stats));
        return img;
    }
    // Camera::debug_ray (src/camera.rs:256-260)
    rt_ray debug_ray(double x, double y) const {
        rt_ray r;
        for (int k = 0; k < 3; ++k) {
            double ps = rec.pixel00[k] + rec.pixel_du[k] * x + rec.pixel_dv[k] * y;
            r.o[k] = rec.center[k]; r.d[k] = ps - rec.center[k];
        }
        return r;
    }
};

}  // namespace rt
