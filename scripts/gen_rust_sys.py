"""Writes rust/concrete-amd-sys/src/lib.rs: the raw Rust declarations of every function that
include/concrete_amd.h and include/concrete_amd_client.h declare. The file is generated so that it
cannot drift from the headers; tests/test_host_cpu.py regenerates it and compares.

    python scripts/gen_rust_sys.py            # rewrite the file
    python scripts/gen_rust_sys.py --check    # exit 1 if the committed file is out of date
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = ["concrete_amd.h", "concrete_amd_client.h"]
OUT = os.path.join(ROOT, "rust", "concrete-amd-sys", "src", "lib.rs")

BASE = {"int": "c_int", "void": "c_void", "char": "c_char", "size_t": "usize", "double": "f64",
        "uint32_t": "u32", "uint64_t": "u64", "camd_context": "CamdContext", "camd_server_key": "CamdServerKey",
        "camd_params": "CamdParams", "camd_rng": "CamdRng"}


def rust_type(c: str) -> str:
    c = c.strip()
    const = c.startswith("const ")
    if const:
        c = c[6:].strip()
    stars = c.count("*")
    base = BASE[c.replace("*", "").strip()]
    if stars == 0:
        return base
    t = base
    for level in range(stars):
        # only the innermost pointee can be const in these headers
        t = ("*const " if (const and level == 0) else "*mut ") + t
    return t


def declarations():
    """[(name, return type, [(argument name, type)])] in header order."""
    out = []
    for h in HEADERS:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
        text = re.sub(r"typedef struct camd_\w+\s*\{.*?\}\s*camd_\w+;", "", text, flags=re.S)
        for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(camd_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
            ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
            params = []
            if args and args != "void":
                for a in args.split(","):
                    mm = re.match(r"(.*?)([A-Za-z_]\w*)$", a.strip())
                    params.append((mm.group(2), rust_type(mm.group(1))))
            out.append((name, rust_type(ret), params))
    return out


PRELUDE = '''//! Raw declarations of `libconcrete_amd.so`, the MI355X (gfx950) keyswitch + programmable-bootstrap
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
'''


def render() -> str:
    lines = [PRELUDE]
    for name, ret, params in declarations():
        args = ", ".join(f"{n}: {t}" for n, t in params)
        tail = "" if ret == "c_void" else f" -> {ret}"
        lines.append(f"    pub fn {name}({args}){tail};\n")
    lines.append("}\n")
    return "".join(lines)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(declarations())} functions")
