// render_small.cpp — compiled-host example above the C ABI using the C++ mirror (rt_amd/host/rt.hpp):
// the reference's usage pattern (src/main.rs:23-61) — camera, checker ground, a few spheres,
// World::build, render_image — ending in a PPM like Camera::write_image (src/camera.rs:343-364).
//   usage: render_small <out.ppm> <out.f64>      (raw little-endian f64 RGB for the parity test)
#include <cstdio>
#include <vector>

#include "../rt_amd/host/rt.hpp"

int main(int argc, char** argv) {
    using namespace rt;
    if (argc < 3) { std::fprintf(stderr, "usage: %s out.ppm out.f64\n", argv[0]); return 2; }
    try {
        Camera cam = Camera::create({14.0, 3.0, 10.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 1.0}, 16.0, 0.7, 96, 54, 4, 50, 20.0);
        auto checker = Lambertian(CheckerTexture(3.0, SolidColor({0.1, 0.1, 0.1}), SolidColor({0.95, 0.95, 0.95})));
        std::vector<Shape> shapes = generate_ground_plane(10000.0, 10000.0, -0.2, checker, true);
        const double fuzz = 0.25;
        shapes.push_back(Sphere({-1.0, -1.732, 0.5}, 0.7, Dielectric(1.5)));
        shapes.push_back(Sphere({-2.0, 0.0, 0.5}, 0.7, Metal(SolidColor({0.7, 0.6, 0.5}))));
        shapes.push_back(Sphere({1.0, 1.0, 0.0}, 0.2, Metal(SolidColor({0.2, 0.8, 0.3}), &fuzz)));
        shapes.push_back(Sphere::new_facing({-1.0, 1.732, 0.5}, 0.7, Lambertian(SolidColor({0.4, 0.2, 0.9})), {14.0, 3.0, 10.0}));
        rt_sky sky;
        // SKY_SYNTH_V1 placeholder coefficients (rt_amd/sky.py) — NOT the Hosek-Wilkie dataset
        const float P[27] = {-0.60f, -0.20f, 0.60f, 0.80f, -1.0f, 0.30f, 0.010f, 0.55f, 0.50f,
                             -0.58f, -0.18f, 0.70f, 0.75f, -1.1f, 0.32f, 0.010f, 0.60f, 0.50f,
                             -0.55f, -0.15f, 0.90f, 0.65f, -1.2f, 0.35f, 0.008f, 0.70f, 0.50f};
        for (int i = 0; i < 27; ++i) sky.params[i] = P[i];
        sky.radiances[0] = 1.6f; sky.radiances[1] = 1.9f; sky.radiances[2] = 2.6f;
        sky.sun[0] = 0.0; sky.sun[1] = 0.0; sky.sun[2] = 1.0;
        World world = World::build(shapes, sky, 0);
        rt_stats st;
        Image img = cam.render_image(world, 7, &st);
        std::vector<uint8_t> rgb8(img.rgb.size());
        check(rt_encode_gamma8(img.rgb.data(), img.rgb.size(), rgb8.data()));
        FILE* f = std::fopen(argv[1], "w");
        if (!f) return 1;
        std::fprintf(f, "P3\n%u %u # width, height\n255 # max color value\n", img.width, img.height);
        for (uint32_t y = 0; y < img.height; ++y)
            for (uint32_t x = 0; x < img.width; ++x) {
                const uint8_t* p = &rgb8[(size_t(y) * img.width + x) * 3];
                std::fprintf(f, "%u %u %u%s", p[0], p[1], p[2], x + 1 == img.width ? "\n" : " ");
            }
        std::fclose(f);
        f = std::fopen(argv[2], "wb");
        if (!f) return 1;
        std::fwrite(img.rgb.data(), sizeof(double), img.rgb.size(), f);
        std::fclose(f);
        std::printf("rendered %ux%u, %llu samples, %llu ray segments\n", img.width, img.height,
                    (unsigned long long)st.samples, (unsigned long long)st.ray_segments);
    } catch (const rt::Error& e) {
        std::fprintf(stderr, "rt error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
