//! What `concrete-boolean` gains beside `bootstrapping/amd.rs` for the `amd` feature: the additions to
//! `engine/mod.rs`, `engine/bootstrapping/mod.rs`, `server_key/mod.rs` and `Cargo.toml`, each where
//! its `cuda` sibling stands (`engine/mod.rs:10-11,31-32,38-39,51-59,74-81`,
//! `bootstrapping/mod.rs:4-8`, `server_key/mod.rs:16-19,40-77,143-181`). Only the added items are
//! written out; nothing of the CPU or CUDA paths changes. The features are exclusive, as `cuda` and
//! the default already are: `amd = ["concrete-amd-sys"]` and
//! `#[cfg(all(feature = "cuda", feature = "amd"))] compile_error!("pick one GPU backend");`.

// ---- engine/bootstrapping/mod.rs ----
#[cfg(feature = "amd")]
mod amd;
#[cfg(feature = "amd")]
pub(crate) use amd::{AmdBootstrapKey, AmdBootstrapper};

// ---- engine/mod.rs ----
#[cfg(feature = "amd")]
use bootstrapping::{AmdBootstrapKey, AmdBootstrapper};

#[cfg(feature = "amd")]
pub(crate) type AmdBooleanEngine = BooleanEngine<AmdBootstrapper>;

thread_local! {
    // one engine per thread, like the other two; the device key is shared (`Arc`), and calls that
    // arrive from several threads at once leave the library as one launch
    #[cfg(feature = "amd")]
    static AMD_ENGINE: RefCell<BooleanEngine<AmdBootstrapper>> = RefCell::new(BooleanEngine::<_>::new());
}

#[cfg(feature = "amd")]
impl WithThreadLocalEngine for AmdBooleanEngine {
    fn with_thread_local_mut<R, F>(func: F) -> R
    where
        F: FnOnce(&mut Self) -> R,
    {
        AMD_ENGINE.with(|engine_cell| func(&mut engine_cell.borrow_mut()))
    }
}

#[cfg(feature = "amd")]
impl BooleanEngine<AmdBootstrapper> {
    /// Device key from the CPU key: Fourier transform of the standard-domain bootstrap key on the GPU,
    /// upload of the keyswitch key and of the scheme's one look-up table.
    pub fn create_server_key(&mut self, cpu_key: &CpuBootstrapKey) -> AmdBootstrapKey {
        self.bootstrapper.new_server_key(cpu_key).unwrap()
    }
}

// ---- server_key/mod.rs ----
#[cfg(feature = "amd")]
use crate::engine::{bootstrapping::AmdBootstrapKey, AmdBooleanEngine};
#[cfg(feature = "amd")]
use std::sync::Arc;

// the struct gains one field, beside `cuda_key`:
//     #[cfg(feature = "amd")]
//     amd_key: Arc<AmdBootstrapKey>,
// and the CPU `mod implementation` is gated by `#[cfg(not(any(feature = "cuda", feature = "amd")))]`.

#[cfg(feature = "amd")]
mod implementation {
    use super::*;

    impl RefFromServerKey for AmdBootstrapKey {
        fn get_ref(server_key: &ServerKey) -> &Self {
            &server_key.amd_key
        }
    }

    // every binary gate of `BinaryBooleanGates` resolves its engine and key through these two types,
    // so and/nand/nor/or/xor/xnor need no edit of their own
    impl DefaultImplementation for ServerKey {
        type Engine = AmdBooleanEngine;
        type BootsrapKey = AmdBootstrapKey;
    }
}

impl ServerKey {
    // `mux` names its engine explicitly (it is not part of `BinaryBooleanGates`); the added arm:
    #[cfg(feature = "amd")]
    pub fn mux(&self, ct_condition: &Ciphertext, ct_then: &Ciphertext, ct_else: &Ciphertext) -> Ciphertext {
        AmdBooleanEngine::with_thread_local_mut(|engine| engine.mux(ct_condition, ct_then, ct_else, &self.amd_key))
    }
}

// `ServerKey::new` and deserialization both end in this conversion, so a key read from disk gets its
// device key the same way a fresh one does; serialization keeps writing the CPU key only.
#[cfg(feature = "amd")]
impl From<CpuBootstrapKey> for ServerKey {
    fn from(cpu_key: CpuBootstrapKey) -> Self {
        let amd_key = AmdBooleanEngine::with_thread_local_mut(|engine| engine.create_server_key(&cpu_key));
        Self { cpu_key, amd_key: Arc::new(amd_key) }
    }
}

// ---- client_key/mod.rs ----
// nothing: the `k = 1` assertion there is the CUDA backend's restriction (`client_key/mod.rs:119-124`)
// and stays under `cfg(feature = "cuda")`.
