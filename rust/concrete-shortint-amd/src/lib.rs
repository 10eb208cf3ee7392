//! MI355X backend of the keyswitch + programmable-bootstrap path of `concrete-shortint`.
//!
//! `AmdServerKey` stands where `ShortintEngine::programmable_bootstrap_keyswitch_assign` and
//! `keyswitch_bootstrap_assign` stand in the reference
//! (`concrete-shortint/src/engine/server_side/mod.rs:131-155` and `:95-118`): same inputs, same
//! in-place result, no degree update. It adds what the reference lacks, a batched form.
//!
//! The device key is made from the STANDARD-domain bootstrap key, which
//! `new_server_key_with_max_degree` holds before it converts it to the Fourier domain and drops
//! it (`server_side/mod.rs:39-50`); the hook that calls `AmdServerKey::from_keys` goes there.
//!
//! Not compiled in the repository that carries it (its build image has no Rust toolchain);
//! `tests/test_host_cpu.py` checks the raw declarations against the C headers.
use std::collections::HashMap;
use std::error::Error;
use std::ffi::CStr;
use std::fmt;
use std::os::raw::c_int;
use std::ptr;
use std::sync::Mutex;

use concrete_amd_sys::*;
use concrete_core::prelude::*;
use concrete_shortint::ciphertext::Ciphertext;
use concrete_shortint::parameters::Parameters;

/// An error reported by the library: status code and the calling thread's message.
#[derive(Debug)]
pub struct AmdError {
    pub code: c_int,
    pub message: String,
}
impl fmt::Display for AmdError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "concrete_amd error {}: {}", self.code, self.message)
    }
}
impl Error for AmdError {}

/// Turns a status code into a `Result`. The library never panics or aborts; the public façade of
/// concrete-shortint unwraps the `EngineResult` exactly as it does for concrete-core errors
/// (`server_key/mod.rs:215,227`).
pub fn check(code: c_int) -> Result<(), Box<dyn Error>> {
    if code == CAMD_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(camd_last_error()) }.to_string_lossy().into_owned();
    Err(Box::new(AmdError { code, message }))
}

/// Identifier of a look-up table that has been uploaded to the GPU.
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub struct LutId(pub u32);

/// Fourier bootstrap key + keyswitch key + look-up tables resident on one MI355X.
pub struct AmdServerKey {
    ctx: *mut CamdContext,
    key: *mut CamdServerKey,
    big_lwe_size: usize,
    /// look-up tables already on the device, by content: the reference rebuilds its accumulator on
    /// every call (`server_side/mod.rs:231`), here equal tables are uploaded once
    luts: Mutex<HashMap<Vec<u64>, LutId>>,
}

// The library serialises and coalesces concurrent calls on one key: rayon workers of
// concrete-integer's `*_parallelized` operations that call `apply_lookup_table` at the same time
// share one launch.
unsafe impl Send for AmdServerKey {}
unsafe impl Sync for AmdServerKey {}

impl AmdServerKey {
    /// Builds the device key on GPU `device` from the standard-domain bootstrap key and the
    /// keyswitch key. The keys are copied; the forward FFT of the bootstrap key runs on the GPU.
    /// Replaces `fft_engine.convert_lwe_bootstrap_key` (`server_side/mod.rs:47-50`).
    pub fn from_keys(
        engine: &mut DefaultEngine,
        bootstrap_key: &LweBootstrapKey64,
        keyswitch_key: &LweKeyswitchKey64,
        parameters: &Parameters,
        device: c_int,
    ) -> Result<Self, Box<dyn Error>> {
        let params = CamdParams {
            lwe_dimension: parameters.lwe_dimension.0 as u32,
            glwe_dimension: parameters.glwe_dimension.0 as u32,
            polynomial_size: parameters.polynomial_size.0 as u32,
            pbs_base_log: parameters.pbs_base_log.0 as u32,
            pbs_level: parameters.pbs_level.0 as u32,
            ks_base_log: parameters.ks_base_log.0 as u32,
            ks_level: parameters.ks_level.0 as u32,
            torus_bits: 64,
        };
        // concrete-core stores both keys as flat Vec<u64>: [n][level][k+1][k+1][N] and
        // [kN][level][n+1], level 1 first: the layouts the C header states
        let bsk_words: Vec<u64> = engine.consume_retrieve_lwe_bootstrap_key(bootstrap_key.clone())?;
        let ksk_words: Vec<u64> = engine.consume_retrieve_lwe_keyswitch_key(keyswitch_key.clone())?;
        let mut ctx = ptr::null_mut();
        check(unsafe { camd_context_create(device, &mut ctx) })?;
        let mut key = ptr::null_mut();
        let rc = unsafe {
            camd_server_key_create(ctx, &params, bsk_words.as_ptr().cast(), ksk_words.as_ptr().cast(), CAMD_PBS_AUTO, &mut key)
        };
        if let Err(e) = check(rc) {
            unsafe { camd_context_destroy(ctx) };
            return Err(e);
        }
        Ok(Self {
            ctx,
            key,
            big_lwe_size: parameters.glwe_dimension.0 * parameters.polynomial_size.0 + 1,
            luts: Mutex::new(HashMap::new()),
        })
    }

    /// Uploads a look-up table (`ServerKey::generate_accumulator`'s trivial GLWE, (k+1)N words)
    /// unless an equal one is already on the device.
    pub fn lut_id_for(&self, engine: &mut DefaultEngine, acc: &GlweCiphertext64) -> Result<LutId, Box<dyn Error>> {
        let words: Vec<u64> = engine.consume_retrieve_glwe_ciphertext(acc.clone())?;
        let mut luts = self.luts.lock().unwrap();
        if let Some(id) = luts.get(&words) {
            return Ok(*id);
        }
        let mut id = 0u32;
        check(unsafe { camd_lut_upload(self.key, words.as_ptr().cast(), &mut id) })?;
        luts.insert(words, LutId(id));
        Ok(LutId(id))
    }

    /// Keyswitch then programmable bootstrap of one ciphertext, in place. The degree is left as
    /// it is: callers set it afterwards, as in the reference (`server_side/mod.rs:210,235`).
    /// Drop-in for `programmable_bootstrap_keyswitch_assign` (`server_side/mod.rs:131-155`).
    pub fn apply_lookup_table(&self, engine: &mut DefaultEngine, ct: &mut Ciphertext, lut: LutId) -> Result<(), Box<dyn Error>> {
        self.apply_lookup_table_batch(engine, std::slice::from_mut(ct), &[lut])
    }

    /// The same for a batch: `cts[i]` is bootstrapped with `luts[i]`. One call, one launch; the
    /// kN+1 words of every ciphertext travel in one contiguous buffer.
    pub fn apply_lookup_table_batch(&self, engine: &mut DefaultEngine, cts: &mut [Ciphertext], luts: &[LutId]) -> Result<(), Box<dyn Error>> {
        assert_eq!(cts.len(), luts.len(), "one look-up table id per ciphertext");
        if cts.is_empty() {
            return Ok(());
        }
        let w = self.big_lwe_size;
        let mut words = vec![0u64; cts.len() * w];
        for (i, ct) in cts.iter().enumerate() {
            let src: Vec<u64> = engine.consume_retrieve_lwe_ciphertext(ct.ct.clone())?;
            assert_eq!(src.len(), w, "ciphertext does not belong to this key");
            words[i * w..(i + 1) * w].copy_from_slice(&src);
        }
        let ids: Vec<u32> = luts.iter().map(|l| l.0).collect();
        // input and output may alias (the header says so)
        check(unsafe { camd_ks_pbs_batch(self.key, words.as_ptr().cast(), ids.as_ptr(), words.as_mut_ptr().cast(), cts.len()) })?;
        for (i, ct) in cts.iter_mut().enumerate() {
            ct.ct = engine.create_lwe_ciphertext_from(words[i * w..(i + 1) * w].to_vec())?;
        }
        Ok(())
    }

    /// `keyswitch_bootstrap_assign` (`server_side/mod.rs:95-118`): the bootstrap with the key's
    /// default table `x % message_modulus`, which the caller uploaded as `default_lut`.
    pub fn keyswitch_bootstrap(&self, engine: &mut DefaultEngine, ct: &mut Ciphertext, default_lut: LutId) -> Result<(), Box<dyn Error>> {
        self.apply_lookup_table(engine, ct, default_lut)
    }

    /// Raw handle, for callers that keep ciphertexts on the GPU and use the `*_device` functions of
    /// `concrete_amd_sys` (asynchronous on a HIP stream) themselves.
    pub fn raw(&self) -> *mut CamdServerKey {
        self.key
    }
}

impl Drop for AmdServerKey {
    fn drop(&mut self) {
        unsafe {
            camd_server_key_destroy(self.key);
            camd_context_destroy(self.ctx);
        }
    }
}
