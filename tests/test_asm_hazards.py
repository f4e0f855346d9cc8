"""The hand-written traversal loop (trace_loop_asm, rt_kernels.hip) pads the gfx940-family data hazards that hardware does not
interlock by hand; tools/asm_hazards.py checks the device assembly of the whole translation unit -- the compiler's code and the
hand-written loops -- against the same rules (CPU-only: hipcc cross-compiles)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("asm_hazards", os.path.join(ROOT, "tools", "asm_hazards.py"))
hz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hz)


def _kernel(body):
    return "_Zkernel:\n" + "\n".join("\t" + l for l in body) + "\n\ts_endpgm\n"


def test_checker_sees_the_hazards_it_is_for():
    ok = ["v_cmp_lt_f32_e64 s[10:11], 0, v1", "v_mul_f32_e32 v2, v3, v4", "s_nop 0", "v_cndmask_b32_e64 v5, v6, v7, s[10:11]",
          "v_cmp_gt_f32_e32 vcc, s35, v0", "s_nop 1", "v_cndmask_b32_e32 v0, v0, v1, vcc",
          "v_sqrt_f32_e32 v1, v0", "s_nop 0", "v_add_u32_e32 v2, -1, v1",
          "v_div_scale_f32 v20, vcc, v17, v16, v17", "v_mul_f32_e32 v21, v20, v19", "v_fma_f32 v22, -v18, v21, v20", "v_fmac_f32_e32 v21, v22, v19",
          "v_fma_f32 v18, -v18, v21, v20", "v_div_fmas_f32 v18, v18, v19, v21",
          "v_readfirstlane_b32 s30, v9", "s_lshl_b32 s31, s30, 6",                     # a SALU read needs no wait
          "v_cmp_eq_u32_e32 vcc, -1, v9", "s_and_saveexec_b64 s[36:37], vcc"]
    assert hz.check(_kernel(ok)) == []
    bad = {"mask read one instruction after the compare": ["v_cmp_lt_f32_e64 s[10:11], 0, v1", "v_mul_f32_e32 v2, v3, v4", "v_cndmask_b32_e64 v5, v6, v7, s[10:11]"],
           "vcc read at once": ["v_cmp_gt_f32_e32 vcc, s35, v0", "v_cndmask_b32_e32 v0, v0, v1, vcc"],
           "scalar operand read too early": ["v_readfirstlane_b32 s30, v9", "v_cmp_ne_u32_e32 vcc, s30, v9"],
           "square root used at once": ["v_sqrt_f32_e32 v1, v0", "v_add_u32_e32 v2, -1, v1"],
           "v_div_fmas three instructions after its vcc": ["v_div_scale_f32 v20, vcc, v17, v16, v17", "v_mul_f32_e32 v21, v20, v19", "v_fma_f32 v22, -v18, v21, v20",
                                                            "v_fmac_f32_e32 v21, v22, v19", "v_div_fmas_f32 v18, v18, v19, v21"]}
    for what, body in bad.items():
        assert len(hz.check(_kernel(body))) == 1, what


def test_render_kernels_have_no_unpadded_hazard():
    text = hz.device_asm()
    loops = len(re.findall(r"^\s*\.Lrt_top\d+:", text, re.M))
    assert loops >= 8 * 4, loops             # eight octants x (primary, primary + views, single-frame, extension ...) hand-written loops were looked at
    findings = hz.check(text)
    assert findings == [], findings[:5]
