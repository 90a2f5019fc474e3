//! rt-hip-sys — `extern "C"` bindings of `librt_hip.so` (include/rt_hip.h, ABI v2) and a thin safe wrapper.
//!
//! SOURCE ONLY: never compiled in this repository's build image (no cargo / rustc there).  Struct layouts mirror
//! the C header field by field; the C side static_asserts the sizes quoted in the `const _` checks below, and
//! `tests/test_abi.py` checks the same numbers for the ctypes mirror.
//!
//! What a maintainer of thabnir/rt routes through it:
//!   World::build(shapes)            -> GpuWorld::build            (rt_scene_create)
//!   Camera::render_image(&World)    -> GpuWorld::render_image     (rt_render)
//!   Camera::render_pixel in the preview loop (src/window.rs:277) -> GpuWorld::render_pass (rt_render_pass)
//!   World::hit / debug_raycast      -> GpuWorld::trace_closest    (rt_trace_closest)
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RT_ABI_VERSION: u32 = 2;
pub const RT_MISS: u32 = 0xFFFF_FFFF;
pub const RT_MAX_SAMPLE_INDEX: u32 = 1 << 20;
pub const RT_NOALLOC_MAX_RAYS: u32 = 1 << 18;
pub const RT_PASS_QUANTISE: u32 = 1;

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_NO_DEVICE: c_int = -2;
pub const RT_ERR_HIP: c_int = -3;
pub const RT_ERR_OOM: c_int = -4;
pub const RT_ERR_LIMIT: c_int = -5;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_IMAGE: u32 = 2;
pub const RT_TEXEL_RGBA8: u32 = 0;
pub const RT_TEXEL_RGBA16: u32 = 1;

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtSphere { pub center: [f64; 3], pub radius: f64, pub face: [f64; 3], pub material: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtTriangle { pub a: [f64; 3], pub b: [f64; 3], pub c: [f64; 3], pub uv: [[f64; 2]; 3], pub n: [f64; 3], pub material: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtMaterial { pub kind: u32, pub texture: u32, pub has_fuzz: u32, pub _pad: u32, pub fuzz: f64, pub ior: f64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtTexture { pub kind: u32, pub even: u32, pub odd: u32, pub width: u32, pub height: u32, pub texel_format: u32,
                       pub texel_offset: u64, pub rgb: [f64; 3], pub inv_scale: f64 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct RtSky { pub params: [f32; 27], pub radiances: [f32; 3], pub sun: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtCamera { pub center: [f64; 3], pub pixel00: [f64; 3], pub pixel_du: [f64; 3], pub pixel_dv: [f64; 3],
                      pub defocus_u: [f64; 3], pub defocus_v: [f64; 3], pub defocus_angle: f64, pub t_min: f64, pub t_max: f64,
                      pub width: u32, pub height: u32, pub max_depth: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtOpts { pub spp: u32, pub sample_offset: u32, pub sample_stride: u32, pub precision: u32, pub seed: u64,
                    pub device: i32, pub max_paths_in_flight: u32, pub collect_stats: u32, pub flags: u32 }
/// `RtOpts::flags`: rt_render_multi exchanges the per-device sums as f32 (RT_OPT_REDUCE_F32)
pub const RT_OPT_REDUCE_F32: u32 = 1;
/// `RtOpts::flags`: rt_render_accumulate_device queues the frame and returns; rt_scene_sync waits (RT_OPT_PIPELINED)
pub const RT_OPT_PIPELINED: u32 = 2;
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtStats {
    pub samples: u64, pub ray_segments: u64, pub nodes_visited: u64, pub prims_tested: u64,
    pub passes: u32, pub launches: u32,
    pub seconds_trace: f64, pub seconds_shade: f64, pub seconds_total: f64,
    pub trace_launches: u64, pub primary_segments: u64, pub seconds_trace_primary: f64,
    pub nodes_visited_primary: u64, pub prims_tested_primary: u64,
    // ABI v2
    pub bounce_segments: u64, pub seconds_trace_bounce: f64, pub bounce_launches: u64,
    pub tail_segments: u64, pub seconds_tail: f64,
    pub nodes_visited_bounce: u64, pub prims_tested_bounce: u64,
    pub wave_node_fetches_primary: u64, pub wave_prim_fetches_primary: u64,
}
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtRay { pub o: [f64; 3], pub d: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtHit { pub t: f64, pub prim: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct RtSceneInfo {
    pub n_prims: u64, pub n_nodes: u64, pub device_bytes: u64,
    pub bvh_depth: u32, pub max_leaf_size: u32, pub node_bytes: u32, pub tri_bytes: u32, pub sphere_bytes: u32, pub bvh_width: u32,
    pub build_seconds: f64, pub upload_seconds: f64, pub builder: u32, pub _pad: u32,
}
pub enum RtScene {}

// sizes the C side static_asserts (rt_amd/csrc/rt_api.hip)
const _: () = assert!(std::mem::size_of::<RtSphere>() == 64);
const _: () = assert!(std::mem::size_of::<RtTriangle>() == 152);
const _: () = assert!(std::mem::size_of::<RtMaterial>() == 32);
const _: () = assert!(std::mem::size_of::<RtTexture>() == 64);
const _: () = assert!(std::mem::size_of::<RtSky>() == 144);
const _: () = assert!(std::mem::size_of::<RtCamera>() == 184);
const _: () = assert!(std::mem::size_of::<RtOpts>() == 40);
const _: () = assert!(std::mem::size_of::<RtStats>() == 176);
const _: () = assert!(std::mem::size_of::<RtRay>() == 48);
const _: () = assert!(std::mem::size_of::<RtHit>() == 16);
const _: () = assert!(std::mem::size_of::<RtSceneInfo>() == 72);

extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count(count: *mut c_int) -> c_int;
    pub fn rt_scene_create(spheres: *const RtSphere, n_spheres: usize, triangles: *const RtTriangle, n_triangles: usize,
                           materials: *const RtMaterial, n_materials: usize, textures: *const RtTexture, n_textures: usize,
                           texel_pool: *const u8, texel_pool_bytes: usize, sky: *const RtSky, device: c_int,
                           out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_scene_get_info(scene: *const RtScene, out: *mut RtSceneInfo) -> c_int;
    pub fn rt_trace_closest(scene: *const RtScene, rays: *const RtRay, n: usize, t_min: f64, t_max: f64, out: *mut RtHit,
                            stats: *mut RtStats) -> c_int;
    pub fn rt_trace_closest_noalloc(scene: *const RtScene, d_rays: *const RtRay, n: usize, t_min: f64, t_max: f64,
                                    d_hits: *mut RtHit, stream: *mut c_void) -> c_int;
    pub fn rt_trace_primary(scene: *const RtScene, cam: *const RtCamera, sample_index: u32, jitter: c_int, seed: u64,
                            out: *mut RtHit, rays_out: *mut RtRay) -> c_int;
    pub fn rt_render(scene: *const RtScene, cam: *const RtCamera, opts: *const RtOpts, out_rgb_mean: *mut f64,
                     stats: *mut RtStats) -> c_int;
    pub fn rt_render_accumulate_device(scene: *const RtScene, cam: *const RtCamera, opts: *const RtOpts, d_rgb_sum: *mut f64,
                                       stream: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_scene_sync(scene: *const RtScene, stats: *mut RtStats) -> c_int;
    pub fn rt_resolve_device(d_rgb_sum: *const f64, d_rgb_mean: *mut f64, n_values: usize, total_spp: u64, device: c_int,
                             stream: *mut c_void) -> c_int;
    pub fn rt_render_multi(scenes: *const *mut RtScene, n_devices: u32, cam: *const RtCamera, opts: *const RtOpts,
                           out_rgb_mean: *mut f64, stats: *mut RtStats) -> c_int;
    pub fn rt_render_pass(scene: *const RtScene, cam: *const RtCamera, n_samples: u32, total_samples: u64, seed: u64,
                          flags: u32, out_rgba8: *mut u8, out_rgb_mean: *mut f64, stats: *mut RtStats) -> c_int;
    pub fn rt_encode_gamma8_device(d_rgb_mean: *const f64, n_values: usize, d_out: *mut u8, device: c_int,
                                   stream: *mut c_void) -> c_int;
    pub fn rt_encode_gamma8(rgb_mean: *const f64, n_values: usize, out: *mut u8) -> c_int;
}

/// An error of the C ABI: the status code and the thread-local message behind it.  Nothing panics or aborts across
/// the boundary; the reference's `expect`s (src/hittable.rs:35, :506, :558, :590) stay on the Rust side.
#[derive(Debug, Clone)]
pub struct RtError { pub code: i32, pub message: String }
impl std::fmt::Display for RtError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result { write!(f, "rt_hip error {}: {}", self.code, self.message) }
}
impl std::error::Error for RtError {}
fn check(code: c_int) -> Result<(), RtError> {
    if code == RT_OK { return Ok(()); }
    let message = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
    Err(RtError { code, message })
}

/// The arrays `rt_scene_create` takes: what `World::build` hands over (flatten.rs fills it from `Vec<Shape>`).
#[derive(Default)]
pub struct FlatScene {
    pub spheres: Vec<RtSphere>, pub triangles: Vec<RtTriangle>, pub materials: Vec<RtMaterial>,
    pub textures: Vec<RtTexture>, pub texel_pool: Vec<u8>, pub sky: Option<RtSky>,
    /// prim id (ABI order: spheres, then triangles) -> index in the original `Vec<Shape>`
    pub prim_to_shape: Vec<u32>,
}

/// `World`'s GPU half: owns the scene handle (immutable after build, `Send + Sync` like `Arc<World>`).
pub struct GpuWorld { scene: *mut RtScene, pub device: i32, pub prim_to_shape: Vec<u32> }
unsafe impl Send for GpuWorld {}
unsafe impl Sync for GpuWorld {}
impl Drop for GpuWorld { fn drop(&mut self) { unsafe { rt_scene_destroy(self.scene) } } }

impl GpuWorld {
    pub fn build(flat: &FlatScene, device: i32) -> Result<Self, RtError> {
        assert_eq!(unsafe { rt_abi_version() }, RT_ABI_VERSION, "librt_hip.so and rt-hip-sys disagree on the ABI version");
        let sky = flat.sky.expect("FlatScene::sky: pass SkyState::raw() and World::sun_direction");
        let mut scene: *mut RtScene = std::ptr::null_mut();
        check(unsafe {
            rt_scene_create(flat.spheres.as_ptr(), flat.spheres.len(), flat.triangles.as_ptr(), flat.triangles.len(),
                            flat.materials.as_ptr(), flat.materials.len(), flat.textures.as_ptr(), flat.textures.len(),
                            flat.texel_pool.as_ptr(), flat.texel_pool.len(), &sky, device, &mut scene)
        })?;
        Ok(GpuWorld { scene, device, prim_to_shape: flat.prim_to_shape.clone() })
    }
    pub fn info(&self) -> Result<RtSceneInfo, RtError> {
        let mut i = RtSceneInfo::default();
        check(unsafe { rt_scene_get_info(self.scene, &mut i) })?;
        Ok(i)
    }
    /// `World::hit` for a batch of rays (src/hittable.rs:137-148): `(index in Vec<Shape>, t)` or None per ray.
    pub fn trace_closest(&self, rays: &[RtRay], t_min: f64, t_max: f64) -> Result<Vec<Option<(usize, f64)>>, RtError> {
        let mut hits = vec![RtHit::default(); rays.len()];
        check(unsafe { rt_trace_closest(self.scene, rays.as_ptr(), rays.len(), t_min, t_max, hits.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(hits.iter().map(|h| if h.prim == RT_MISS { None } else {
            Some((self.prim_to_shape.get(h.prim as usize).copied().unwrap_or(h.prim) as usize, h.t)) }).collect())
    }
    /// `Camera::render_image` (src/camera.rs:327-341): linear f64 RGB means, `out[(y * width + x) * 3 + c]`.
    pub fn render_image(&self, cam: &RtCamera, spp: u32, seed: u64) -> Result<(Vec<f64>, RtStats), RtError> {
        let opts = RtOpts { spp, sample_offset: 0, sample_stride: 1, precision: 64, seed, device: self.device, ..Default::default() };
        let mut out = vec![0.0f64; cam.width as usize * cam.height as usize * 3];
        let mut stats = RtStats::default();
        check(unsafe { rt_render(self.scene, cam, &opts, out.as_mut_ptr(), &mut stats) })?;
        Ok((out, stats))
    }
    /// One sweep of `window::render_thread` (src/window.rs:256-311): returns the 8-bit linear RGBA frame for `pixels`.
    pub fn render_pass(&self, cam: &RtCamera, n_samples: u32, total_samples: u64, seed: u64, quantise: bool,
                       frame_rgba8: &mut [u8]) -> Result<RtStats, RtError> {
        assert_eq!(frame_rgba8.len(), cam.width as usize * cam.height as usize * 4);
        let mut stats = RtStats::default();
        check(unsafe { rt_render_pass(self.scene, cam, n_samples, total_samples, seed, if quantise { RT_PASS_QUANTISE } else { 0 },
                                      frame_rgba8.as_mut_ptr(), std::ptr::null_mut(), &mut stats) })?;
        Ok(stats)
    }
    pub fn raw(&self) -> *mut RtScene { self.scene }
}

/// `rt_render_multi`: the same scene built once per GPU; samples partitioned by index, RCCL sum-reduce over xGMI.
pub fn render_image_multi(worlds: &[&GpuWorld], cam: &RtCamera, spp_total: u32, seed: u64) -> Result<Vec<f64>, RtError> {
    let handles: Vec<*mut RtScene> = worlds.iter().map(|w| w.raw()).collect();
    let opts = RtOpts { spp: spp_total, sample_offset: 0, sample_stride: 1, precision: 64, seed, device: worlds[0].device, ..Default::default() };
    let mut out = vec![0.0f64; cam.width as usize * cam.height as usize * 3];
    check(unsafe { rt_render_multi(handles.as_ptr(), handles.len() as u32, cam, &opts, out.as_mut_ptr(), std::ptr::null_mut()) })?;
    Ok(out)
}

#[cfg(feature = "flatten")]
pub mod flatten;
