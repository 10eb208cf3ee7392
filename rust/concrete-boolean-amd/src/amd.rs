//! A third `Bootstrapper` for concrete-boolean, next to `cpu.rs` and `cuda.rs`: this file goes to
//! `concrete-boolean/src/engine/bootstrapping/amd.rs` (the trait is crate-private,
//! `bootstrapping/mod.rs:22-48`) behind `#[cfg(feature = "amd")]`, selected where the CUDA one is
//! (`server_key/mod.rs:47-77`). Structure follows `cuda.rs:140-238`; the differences: no per-call
//! device buffers (scratch lives in the device key), no `k = 1` restriction
//! (`client_key/mod.rs:119-124`), and concurrent calls on one key share a launch.
use super::{BooleanServerKey, Bootstrapper, CpuBootstrapKey};
use crate::ciphertext::Ciphertext;
use crate::PLAINTEXT_TRUE;
use concrete_amd_sys::*;
use concrete_core::prelude::*;
use std::error::Error;
use std::ffi::CStr;
use std::ptr;

fn check(code: std::os::raw::c_int) -> Result<(), Box<dyn Error>> {
    if code == CAMD_OK {
        return Ok(());
    }
    Err(unsafe { CStr::from_ptr(camd_last_error()) }.to_string_lossy().into_owned().into())
}

pub(crate) struct AmdBootstrapKey {
    ctx: *mut CamdContext,
    key: *mut CamdServerKey,
    lwe_size: LweSize,
    big_lwe_size: LweSize,
}
unsafe impl Send for AmdBootstrapKey {}
unsafe impl Sync for AmdBootstrapKey {}

impl BooleanServerKey for AmdBootstrapKey {
    fn lwe_size(&self) -> LweSize {
        self.lwe_size
    }
}

impl Drop for AmdBootstrapKey {
    fn drop(&mut self) {
        unsafe {
            camd_server_key_destroy(self.key);
            camd_context_destroy(self.ctx);
        }
    }
}

pub(crate) struct AmdBootstrapper {
    // Secret does not matter, no key or ciphertext is generated with it (as in cuda.rs:113-114)
    cpu_engine: DefaultEngine,
}

impl Default for AmdBootstrapper {
    fn default() -> Self {
        Self { cpu_engine: DefaultEngine::new(Box::new(UnixSeeder::new(0))).unwrap() }
    }
}

impl AmdBootstrapper {
    /// Device key from the CPU key's standard-domain bootstrap key (`cuda.rs:121-137`), 32-bit
    /// torus. The one look-up table of the boolean scheme (body = 1/8, `cpu.rs:49-53`) is uploaded here.
    pub(crate) fn new_server_key(&mut self, server_key: &CpuBootstrapKey) -> Result<AmdBootstrapKey, Box<dyn Error>> {
        let bsk = &server_key.standard_bootstraping_key;
        let ksk = &server_key.key_switching_key;
        let params = CamdParams {
            lwe_dimension: bsk.input_lwe_dimension().0 as u32,
            glwe_dimension: bsk.glwe_dimension().0 as u32,
            polynomial_size: bsk.polynomial_size().0 as u32,
            pbs_base_log: bsk.decomposition_base_log().0 as u32,
            pbs_level: bsk.decomposition_level_count().0 as u32,
            ks_base_log: ksk.decomposition_base_log().0 as u32,
            ks_level: ksk.decomposition_level_count().0 as u32,
            torus_bits: 32,
        };
        let bsk_words: Vec<u32> = self.cpu_engine.consume_retrieve_lwe_bootstrap_key(bsk.clone())?;
        let ksk_words: Vec<u32> = self.cpu_engine.consume_retrieve_lwe_keyswitch_key(ksk.clone())?;
        let mut ctx = ptr::null_mut();
        check(unsafe { camd_context_create(0, &mut ctx) })?;
        let mut key = ptr::null_mut();
        if let Err(e) = check(unsafe {
            camd_server_key_create(ctx, &params, bsk_words.as_ptr().cast(), ksk_words.as_ptr().cast(), CAMD_PBS_AUTO, &mut key)
        }) {
            unsafe { camd_context_destroy(ctx) };
            return Err(e);
        }
        let glwe_words = (params.glwe_dimension as usize + 1) * params.polynomial_size as usize;
        let mut acc = vec![0u32; glwe_words];
        acc[glwe_words - params.polynomial_size as usize..].fill(PLAINTEXT_TRUE);
        let mut lut_id = 0u32;
        let out = AmdBootstrapKey {
            ctx,
            key,
            lwe_size: bsk.input_lwe_dimension().to_lwe_size(),
            big_lwe_size: bsk.output_lwe_dimension().to_lwe_size(),
        };
        check(unsafe { camd_lut_upload(key, acc.as_ptr().cast(), &mut lut_id) })?;
        Ok(out)
    }
}

impl Bootstrapper for AmdBootstrapper {
    type ServerKey = AmdBootstrapKey;

    fn bootstrap(&mut self, input: &LweCiphertext32, server_key: &Self::ServerKey) -> Result<LweCiphertext32, Box<dyn Error>> {
        let words: Vec<u32> = self.cpu_engine.consume_retrieve_lwe_ciphertext(input.clone())?;
        let mut out = vec![0u32; server_key.big_lwe_size.0];
        check(unsafe { camd_pbs_batch(server_key.key, words.as_ptr().cast(), ptr::null(), out.as_mut_ptr().cast(), 1) })?;
        Ok(self.cpu_engine.create_lwe_ciphertext_from(out)?)
    }

    fn keyswitch(&mut self, input: &LweCiphertext32, server_key: &Self::ServerKey) -> Result<LweCiphertext32, Box<dyn Error>> {
        let words: Vec<u32> = self.cpu_engine.consume_retrieve_lwe_ciphertext(input.clone())?;
        let mut out = vec![0u32; server_key.lwe_size.0];
        check(unsafe { camd_ks_batch(server_key.key, words.as_ptr().cast(), out.as_mut_ptr().cast(), 1) })?;
        Ok(self.cpu_engine.create_lwe_ciphertext_from(out)?)
    }

    fn bootstrap_keyswitch(&mut self, ciphertext: LweCiphertext32, server_key: &Self::ServerKey) -> Result<Ciphertext, Box<dyn Error>> {
        // the input's storage is refilled with the result, as cpu.rs:245-256 and cuda.rs:226-237 do
        let mut data: Vec<u32> = self.cpu_engine.consume_retrieve_lwe_ciphertext(ciphertext)?;
        let p = data.as_mut_ptr();
        check(unsafe { camd_pbs_ks_batch(server_key.key, p.cast_const().cast(), ptr::null(), p.cast(), 1) })?;
        Ok(Ciphertext::Encrypted(self.cpu_engine.create_lwe_ciphertext_from(data)?))
    }
}
