//! dump_render — render dump of the reference WITH integration/rust/patches/rt-counter-rng.patch applied, for
//! pinning the whole render path (camera rays, scatter, roulette, sky) and not only `World::hit`.
//!
//! Not compiled in this repository's image (no cargo/rustc there): source only.  In the patched reference checkout,
//! as `examples/dump_render.rs`:
//!     cargo run --release --example dump_render -- scene.rthx render.rtrd \
//!         W H SPP MAX_DEPTH SEED  cx cy cz  lx ly lz  VFOV FOCUS_DIST DEFOCUS_ANGLE
//! then, on a machine with the library built,
//!     python -m rt_amd.hits_exchange compare-render scene.rthx render.rtrd
//! which renders the same scene, camera, seed and sky state through `rt_render` and compares the mean radiance
//! per pixel to rtol 2e-6: the sky's f32 cos/exp/pow come from different math libraries on the two sides, and the
//! reference adds a pixel's samples with a rayon `.sum()` (src/camera.rs:323), whose association order is not fixed.
//!
//! Scene: the shapes of the .rthx file (formats in dump_hits.rs) with one grey Lambertian (0.5, 0.5, 0.5) — the same
//! stand-in `hits_exchange.desc_from_exchange` builds.  The sky is the reference's own: `World::build` makes the
//! `SkyState`, and its `raw()` parameters travel in the dump so the library evaluates the very same coefficients.
//!
//! .rtrd  "RTRD0001" | u32 W | u32 H | u32 spp | u32 max_depth | u64 seed
//!        | 12 f64 (center xyz, lookat xyz, up xyz, vfov, focus_distance, defocus_angle)
//!        | 27 f32 sky params | 3 f32 sky radiances | 3 f64 sun direction
//!        | W*H*3 f64 mean radiance, row-major, RGB
use rt::{
    camera::{Camera, Float},
    hittable::{Shape, Sphere, Triangle, World},
    material::{Lambertian, Material},
    vec3::{Vec2, Vec3},
};
use std::{
    convert::TryInto,
    fs,
    io::{BufWriter, Write},
    sync::Arc,
};

struct Reader<'a> {
    buf: &'a [u8],
    pos: usize,
}
impl<'a> Reader<'a> {
    fn bytes(&mut self, n: usize) -> &'a [u8] {
        let s = &self.buf[self.pos..self.pos + n];
        self.pos += n;
        s
    }
    fn u64(&mut self) -> u64 {
        u64::from_le_bytes(self.bytes(8).try_into().unwrap())
    }
    fn f64(&mut self) -> Float {
        Float::from_le_bytes(self.bytes(8).try_into().unwrap())
    }
    fn vec3(&mut self) -> Vec3 {
        let (x, y, z) = (self.f64(), self.f64(), self.f64());
        Vec3::new(x, y, z)
    }
    fn vec2(&mut self) -> Vec2 {
        let (x, y) = (self.f64(), self.f64());
        Vec2::new(x, y)
    }
}

fn main() {
    let a: Vec<String> = std::env::args().collect();
    if a.len() != 17 {
        eprintln!("usage: dump_render scene.rthx out.rtrd W H SPP MAX_DEPTH SEED cx cy cz lx ly lz VFOV FOCUS DEFOCUS");
        std::process::exit(2);
    }
    let num = |i: usize| -> Float { a[i].parse().expect("numeric argument") };
    let (w, h, spp, max_depth) = (num(3) as usize, num(4) as usize, num(5) as usize, num(6) as usize);
    let seed: u64 = a[7].parse().expect("seed");
    let center = Vec3::new(num(8), num(9), num(10));
    let lookat = Vec3::new(num(11), num(12), num(13));
    let up = Vec3::new(0.0, 1.0, 0.0);
    let (vfov, focus, defocus) = (num(14), num(15), num(16));

    let data = fs::read(&a[1]).expect("cannot read the exchange file");
    let mut r = Reader { buf: &data, pos: 0 };
    assert_eq!(r.bytes(8), b"RTHX0001", "not an RTHX0001 file");
    let (n_spheres, n_triangles, _n_rays) = (r.u64() as usize, r.u64() as usize, r.u64() as usize);
    let (_t_min, _t_max) = (r.f64(), r.f64());
    let material: Arc<Material> = Arc::new(Lambertian::new_rgb_solid(0.5, 0.5, 0.5).into());
    let mut shapes: Vec<Shape> = Vec::with_capacity(n_spheres + n_triangles);
    for _ in 0..n_spheres {
        let c = r.vec3();
        let radius = r.f64();
        let facing = r.vec3();
        shapes.push(Sphere::new_facing(c, radius, material.clone(), facing).into());
    }
    for _ in 0..n_triangles {
        let (p, q, s) = (r.vec3(), r.vec3(), r.vec3());
        let (uv_a, uv_b, uv_c) = (r.vec2(), r.vec2(), r.vec2());
        shapes.push(Triangle::new_with_uv(p, q, s, uv_a, uv_b, uv_c, material.clone()).into());
    }
    let world = World::build(shapes);

    let mut camera = Camera::new(center, lookat, up, focus, defocus, w, h, spp, max_depth, vfov, 0.0..Float::MAX);
    camera.seed = seed; // field added by rt-counter-rng.patch
    let image = camera.render_image(&world);

    let mut out = BufWriter::new(fs::File::create(&a[2]).expect("cannot create the dump file"));
    out.write_all(b"RTRD0001").unwrap();
    for v in [w as u32, h as u32, spp as u32, max_depth as u32] {
        out.write_all(&v.to_le_bytes()).unwrap();
    }
    out.write_all(&seed.to_le_bytes()).unwrap();
    for v in [center.x, center.y, center.z, lookat.x, lookat.y, lookat.z, up.x, up.y, up.z, vfov, focus, defocus] {
        out.write_all(&v.to_le_bytes()).unwrap();
    }
    let (params, radiances) = world.sky.raw(); // hw_skymodel::rgb::SkyState::raw -> ([f32; 27], [f32; 3])
    for p in params.iter().chain(radiances.iter()) {
        out.write_all(&p.to_le_bytes()).unwrap();
    }
    for v in [world.sun_direction.x, world.sun_direction.y, world.sun_direction.z] {
        out.write_all(&v.to_le_bytes()).unwrap();
    }
    // render_image collects (x, y, colour) in row-major order of (y, x) (src/camera.rs:328-334)
    let mut means = vec![0.0 as Float; w * h * 3];
    for (x, y, c) in image.pixels.iter() {
        let i = (y * w + x) * 3;
        means[i] = c.x;
        means[i + 1] = c.y;
        means[i + 2] = c.z;
    }
    for v in means {
        out.write_all(&v.to_le_bytes()).unwrap();
    }
    out.flush().unwrap();
    eprintln!("{}x{} at {} spp, depth {}, seed {} -> {}", w, h, spp, max_depth, seed, a[2]);
}
