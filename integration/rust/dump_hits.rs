//! dump_hits — closest-hit dump of the UNMODIFIED reference for the parity exchange file.
//!
//! Not compiled in this repository's image (no cargo/rustc there): source only.  Drop it into the reference
//! checkout as `examples/dump_hits.rs` and run
//!     cargo run --release --example dump_hits -- scene.rthx hits.rthd
//! then, on a machine with the library built,
//!     python -m rt_amd.hits_exchange compare scene.rthx hits.rthd
//! which runs the same rays through `rt_trace_closest` and compares `(prim, t.to_bits())` record by record.
//!
//! It needs no patch to the reference: `World::shapes` and `World::bvh` are public (src/hittable.rs:24-26), so
//! the loop of `World::hit` (src/hittable.rs:137-148) is repeated here with the candidate's position in
//! `shapes` kept next to its distance, and checked against `World::hit` itself for every ray.
//!
//! File formats (little-endian, written/read by rt_amd/hits_exchange.py):
//!   .rthx  "RTHX0001" | u64 n_spheres | u64 n_triangles | u64 n_rays | f64 t_min | f64 t_max
//!          | n_spheres  x 7 f64  (center xyz, radius, front_direction xyz)
//!          | n_triangles x 15 f64 (a xyz, b xyz, c xyz, uv_a, uv_b, uv_c)
//!          | n_rays x 6 f64      (origin xyz, direction xyz — un-normalised, as handed to Ray::new)
//!   .rthd  "RTHD0001" | u64 n_rays | n_rays x (u32 prim, u32 0, u64 t bits); miss = prim 0xFFFFFFFF, t bits 0
//! Primitive ids: position in `Vec<Shape>` = spheres first, then triangles, in file order.

use rt::{
    camera::Float,
    hittable::{Hit, Shape, Sphere, Triangle, World},
    material::{Lambertian, Material},
    vec3::{Ray, Vec2, Vec3},
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
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: dump_hits <scene.rthx> <hits.rthd>");
        std::process::exit(2);
    }
    let data = fs::read(&args[1]).expect("cannot read the exchange file");
    let mut r = Reader { buf: &data, pos: 0 };
    assert_eq!(r.bytes(8), b"RTHX0001", "not an RTHX0001 file");
    let (n_spheres, n_triangles, n_rays) = (r.u64() as usize, r.u64() as usize, r.u64() as usize);
    let (t_min, t_max) = (r.f64(), r.f64());

    // Closest hits do not depend on the material; one shared Lambertian keeps the constructors' signatures.
    let material: Arc<Material> = Arc::new(Lambertian::new_rgb_solid(0.5, 0.5, 0.5).into());
    let mut shapes: Vec<Shape> = Vec::with_capacity(n_spheres + n_triangles);
    for _ in 0..n_spheres {
        let center = r.vec3();
        let radius = r.f64();
        let facing = r.vec3();
        shapes.push(Sphere::new_facing(center, radius, material.clone(), facing).into());
    }
    for _ in 0..n_triangles {
        let (a, b, c) = (r.vec3(), r.vec3(), r.vec3());
        let (uv_a, uv_b, uv_c) = (r.vec2(), r.vec2(), r.vec2());
        shapes.push(Triangle::new_with_uv(a, b, c, uv_a, uv_b, uv_c, material.clone()).into());
    }
    let world = World::build(shapes);
    let base = world.shapes.as_ptr() as usize;
    let stride = std::mem::size_of::<Shape>();

    let mut out = BufWriter::new(fs::File::create(&args[2]).expect("cannot create the dump file"));
    out.write_all(b"RTHD0001").unwrap();
    out.write_all(&(n_rays as u64).to_le_bytes()).unwrap();
    for _ in 0..n_rays {
        let origin = r.vec3();
        let direction = r.vec3();
        let ray = Ray::new(origin.into(), direction);

        // World::hit with the winning shape's index kept (src/hittable.rs:137-148)
        let mut nearest = t_max;
        let mut prim: u32 = 0xFFFF_FFFF;
        for shape in world.bvh.nearest_traverse_iterator(&ray, &world.shapes) {
            if let Some(hit) = shape.hit(&ray, &(t_min..nearest)) {
                nearest = hit.t;
                prim = ((shape as *const Shape as usize - base) / stride) as u32;
            }
        }
        // the loop above must agree with the reference's own entry point
        match world.hit(&ray, &(t_min..t_max)) {
            Some(hit) => assert!(prim != 0xFFFF_FFFF && hit.t.to_bits() == nearest.to_bits()),
            None => assert!(prim == 0xFFFF_FFFF),
        }
        let t_bits: u64 = if prim == 0xFFFF_FFFF { 0 } else { nearest.to_bits() };
        out.write_all(&prim.to_le_bytes()).unwrap();
        out.write_all(&0u32.to_le_bytes()).unwrap();
        out.write_all(&t_bits.to_le_bytes()).unwrap();
    }
    out.flush().unwrap();
    eprintln!("{} spheres, {} triangles, {} rays -> {}", n_spheres, n_triangles, n_rays, args[2]);
}
