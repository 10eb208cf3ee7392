//! Raw declarations of `libconcrete_amd.so`, the MI355X (gfx950) keyswitch + programmable-bootstrap
//! backend. GENERATED from `include/concrete_amd.h` and `include/concrete_amd_client.h` by
//! `scripts/gen_rust_sys.py`; the headers document every function and name the interface of the
//! reference each one replaces. Safe wrappers live in `concrete-shortint-amd` and
//! `concrete-boolean-amd`.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const CAMD_OK: c_int = 0;
pub const CAMD_EINVAL: c_int = -1;
pub const CAMD_EHIP: c_int = -2;
pub const CAMD_ENOMEM: c_int = -3;
pub const CAMD_ENODEV: c_int = -4;

pub const CAMD_PBS_AUTO: c_int = 0;
pub const CAMD_PBS_GENERIC: c_int = 1;
pub const CAMD_PBS_WAVE: c_int = 2;
pub const CAMD_PBS_BLOCK: c_int = 3;

/// `camd_params`: the cryptographic shape of a server key.
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct CamdParams {
    pub lwe_dimension: u32,
    pub glwe_dimension: u32,
    pub polynomial_size: u32,
    pub pbs_base_log: u32,
    pub pbs_level: u32,
    pub ks_base_log: u32,
    pub ks_level: u32,
    pub torus_bits: u32,
}

/// `camd_rng`: generator state of the host-side client operations.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct CamdRng {
    pub s: [u64; 4],
    pub counter: u64,
    pub secure: u32,
    pub have: u32,
    pub block: [u64; 8],
}

#[repr(C)]
pub struct CamdContext {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct CamdServerKey {
    _opaque: [u8; 0],
}

extern "C" {
    pub fn camd_last_error() -> *const c_char;
    pub fn camd_version() -> *const c_char;
    pub fn camd_context_create(device_id: c_int, out: *mut *mut CamdContext) -> c_int;
    pub fn camd_context_destroy(ctx: *mut CamdContext) -> c_int;
    pub fn camd_context_device(ctx: *const CamdContext, device_id: *mut c_int) -> c_int;
    pub fn camd_context_synchronize(ctx: *mut CamdContext) -> c_int;
    pub fn camd_server_key_create(ctx: *mut CamdContext, params: *const CamdParams, bsk_standard: *const c_void, ksk: *const c_void, pbs_variant: c_int, out: *mut *mut CamdServerKey) -> c_int;
    pub fn camd_server_key_destroy(key: *mut CamdServerKey) -> c_int;
    pub fn camd_server_key_params(key: *const CamdServerKey, out: *mut CamdParams) -> c_int;
    pub fn camd_server_key_variant(key: *const CamdServerKey, variant: *mut c_int) -> c_int;
    pub fn camd_server_key_blob_size(key: *const CamdServerKey, bytes: *mut usize) -> c_int;
    pub fn camd_fourier_layout_id(params: *const CamdParams, pbs_variant: c_int, id: *mut u64) -> c_int;
    pub fn camd_server_key_export_blob(key: *const CamdServerKey, device_dst: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn camd_server_key_import_blob(ctx: *mut CamdContext, params: *const CamdParams, pbs_variant: c_int, device_src: *const c_void, bytes: usize, stream: *mut c_void, out: *mut *mut CamdServerKey) -> c_int;
    pub fn camd_lut_upload(key: *mut CamdServerKey, glwe_host: *const c_void, lut_id: *mut u32) -> c_int;
    pub fn camd_lut_count(key: *const CamdServerKey, count: *mut u32) -> c_int;
    pub fn camd_ks_pbs_batch(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize) -> c_int;
    pub fn camd_pbs_ks_batch(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize) -> c_int;
    pub fn camd_pbs_batch(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize) -> c_int;
    pub fn camd_ks_batch(key: *mut CamdServerKey, ct_in: *const c_void, ct_out: *mut c_void, count: usize) -> c_int;
    pub fn camd_reserve(key: *mut CamdServerKey, count: usize) -> c_int;
    pub fn camd_server_key_check(key: *mut CamdServerKey, stream: *mut c_void) -> c_int;
    pub fn camd_ks_pbs_batch_device(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_ks_pbs_batch_device_noalloc(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_pbs_ks_batch_device(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_pbs_batch_device(key: *mut CamdServerKey, ct_in: *const c_void, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_ks_batch_device(key: *mut CamdServerKey, ct_in: *const c_void, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_lwe_axpby_device(ctx: *mut CamdContext, torus_bits: u32, out: *mut c_void, out_stride: usize, x: *const c_void, x_stride: usize, alpha: u64, y: *const c_void, y_stride: usize, beta: u64, body_add: u64, words: usize, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_axpby_ks_pbs_batch_device(key: *mut CamdServerKey, x: *const c_void, x_stride: usize, alpha: u64, y: *const c_void, y_stride: usize, beta: u64, body_add: u64, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_axpby_pbs_ks_batch_device(key: *mut CamdServerKey, x: *const c_void, x_stride: usize, alpha: u64, y: *const c_void, y_stride: usize, beta: u64, body_add: u64, lut_ids: *const u32, ct_out: *mut c_void, count: usize, stream: *mut c_void) -> c_int;
    pub fn camd_timing_enable(key: *mut CamdServerKey, enable: c_int) -> c_int;
    pub fn camd_measure_fp64_fma_rate(ctx: *mut CamdContext, tflops: *mut f64) -> c_int;
    pub fn camd_timing_read(key: *mut CamdServerKey, pbs_ms_avg: *mut f64, pbs_launches: *mut u64, ks_ms_avg: *mut f64, ks_launches: *mut u64) -> c_int;
    pub fn camd_rng_seed(rng: *mut CamdRng, seed: u64);
    pub fn camd_rng_seed_secure(rng: *mut CamdRng) -> c_int;
    pub fn camd_client_gen_binary_key(rng: *mut CamdRng, torus_bits: u32, len: usize, key: *mut c_void) -> c_int;
    pub fn camd_client_gen_bootstrap_key(rng: *mut CamdRng, params: *const CamdParams, glwe_std_dev: f64, lwe_key: *const c_void, glwe_key: *const c_void, bsk_out: *mut c_void, threads: c_int) -> c_int;
    pub fn camd_client_gen_keyswitch_key(rng: *mut CamdRng, torus_bits: u32, in_key: *const c_void, in_dim: usize, out_key: *const c_void, out_dim: usize, base_log: u32, level: u32, lwe_std_dev: f64, ksk_out: *mut c_void) -> c_int;
    pub fn camd_client_lwe_encrypt(rng: *mut CamdRng, torus_bits: u32, key: *const c_void, dim: usize, plaintexts: *const c_void, count: usize, std_dev: f64, ct_out: *mut c_void) -> c_int;
    pub fn camd_client_lwe_phase(torus_bits: u32, key: *const c_void, dim: usize, ct: *const c_void, count: usize, phases_out: *mut c_void) -> c_int;
    pub fn camd_client_shortint_accumulator(glwe_dimension: u32, polynomial_size: u32, message_modulus: u32, carry_modulus: u32, f_values: *const u64, glwe_out: *mut u64) -> c_int;
    pub fn camd_client_boolean_accumulator(glwe_dimension: u32, polynomial_size: u32, glwe_out: *mut u32) -> c_int;
}
