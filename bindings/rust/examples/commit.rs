//! How nova-snark's commitment call site would use the resident-generator path: the
//! generators of `CommitGens` are uploaded once with their fixed-base table, and `commit` becomes one `vdf_msm`.
//! Sketch only -- never compiled in the build image (no Rust toolchain).  `pallas::Affine`, `pallas::Point` and
//! `pallas::Scalar` of pasta_curves with the `repr-c` feature have the layouts of VdfAffine, VdfJac and VdfFe.
use std::ptr;
use vdf_hip_sys::*;

pub struct GpuCommitGens {
    ctx: *mut VdfCtx,
    bases: *mut VdfBases,
}

fn check(ctx: *mut VdfCtx, rc: i32) -> Result<(), String> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(vdf_last_error(ctx)) };
    Err(format!("vdf error {}: {}", rc, msg.to_string_lossy()))
}

impl GpuCommitGens {
    /// `PublicParams::setup` (src/nova/proof.rs:236): after nova-snark has derived its generators.
    pub fn new(gens: &[VdfAffine]) -> Result<Self, String> {
        let mut ctx = ptr::null_mut();
        let dev = 0i32;
        check(ptr::null_mut(), unsafe { vdf_ctx_create(&dev, 1, &mut ctx) })?;
        let mut bases = ptr::null_mut();
        check(ctx, unsafe { vdf_bases_upload(ctx, 0 /* VDF_CURVE_PALLAS */, gens.as_ptr(), gens.len(), &mut bases) })?;
        check(ctx, unsafe { vdf_bases_precompute(ctx, bases, 16, 1) })?;
        Ok(GpuCommitGens { ctx, bases })
    }

    /// nova-snark `commit(gens, v)` -> pasta-msm `vartime_multiscalar_mul`: scalars are Montgomery in memory.
    pub fn commit(&self, v: &[VdfFe]) -> Result<VdfJac, String> {
        let mut out = VdfJac { x: VdfFe { l: [0; 4] }, y: VdfFe { l: [0; 4] }, z: VdfFe { l: [0; 4] } };
        check(self.ctx, unsafe { vdf_msm(self.ctx, self.bases, 0, v.as_ptr(), v.len(), 1, &mut out) })?;
        Ok(out)
    }
}

impl Drop for GpuCommitGens {
    fn drop(&mut self) {
        unsafe {
            vdf_bases_free(self.bases);
            vdf_ctx_destroy(self.ctx);
        }
    }
}
