"""profiles/traffic_<cfg>.json from a tools/pmc.sh summary: HBM bytes per launch per kernel =
2 * FETCH_SIZE * 1024 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md section HBM; cross-checked
here with TCC_EA0_RDREQ_128B * 128) + WRITE_SIZE * 1024."""
import hashlib, json, os, re, sys
summary, cfg, B, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur, vals, names = None, {}, []
for line in open(summary):
    if not line.startswith(" "):
        name = line.strip()
        names.append(name)
        cur = name.split("<")[0]
        targs = [a.strip() for a in name[name.find("<") + 1:name.rfind(">")].split(",")] if "<" in name else []
        # the factored-z instantiations are kept apart: template argument ZF = the last one of k_grads_x<KP, HASA, TERMS, ZF> and
        # k_moments_x<KP, PREDICT, NW, ZF>, the third of k_grads_t<KP, HASA, ZF, IDX>
        zf = (cur in ("k_grads_x", "k_moments_x") and targs and targs[-1] == "true") or (cur in ("k_grads_t", "k_s12_x", "k_grads") and len(targs) >= 3 and targs[2] == "true")
        if zf:
            cur += "_zfac"
        if cur == "k_moments_x" and ", true, 4" in name:      # the prediction instantiation of pass 1
            cur = "k_moments_x_predict"
        vals.setdefault(cur, {})
    else:
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+(\S+)", line)
        if m: vals[cur][m.group(1)] = float(m.group(2))
lib = os.path.join(REPO, "qfa_amd", "libqfa_hip.so")
res = {"config": cfg, "B": B,
       # what was measured: bench.py compares this hash with the library it loaded and says `traffic_stale` when they differ
       "libqfa_hip_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,
       "kernels_measured": names,
       "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ*); "
       "bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; check: TCC_EA0_RDREQ_128B*128"}
# (k_prep_pst: the operand images of the pixel-resident pass 2, written behind the solve and counted with it)
if "k_prep_pst" in vals and "k_solve" in vals:
    for c in vals["k_prep_pst"]: vals["k_solve"][c] = vals["k_solve"].get(c, 0) + vals["k_prep_pst"][c]
for k in ("k_grads", "k_grads_x", "k_grads_t", "k_moments", "k_solve", "k_grads_x_zfac", "k_grads_t_zfac", "k_moments_x_zfac"):
    v = vals.get(k, {})
    if k == "k_moments" and "k_moments_x" in vals:      # pass 1 on the XDL pipe (N_h <= 16)
        v = vals["k_moments_x"]
        res["k_moments_kernel"] = "k_moments_x"
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        res[k + "_hbm_bytes_per_launch"] = 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
        res[k + "_read_bytes_rdreq128"] = v.get("TCC_EA0_RDREQ_128B", 0) * 128 + v.get("TCC_EA0_RDREQ_64B", 0) * 64
        res[k + "_write_bytes"] = v["WRITE_SIZE"] * 1024
# N_h = 17..32: pass 2 is three launches (k_s12_x, then k_grads_s3 once per 16 output columns: its mean per launch counts twice);
# bench.py names the sum "k_s12_x+2*k_grads_s3"
for zs in ("", "_zfac"):
    a, b = vals.get("k_s12_x" + zs, {}), vals.get("k_grads_s3", {})
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a and "FETCH_SIZE" in b and "WRITE_SIZE" in b:
        res["k_s12_x+2*k_grads_s3" + zs + "_hbm_bytes_per_launch"] = (2 * a["FETCH_SIZE"] + a["WRITE_SIZE"] + 2 * (2 * b["FETCH_SIZE"] + b["WRITE_SIZE"])) * 1024
for k in ("k_predict_x", "k_predict_x32", "k_moments_x_predict", "k_zfactor_check"):
    v = vals.get(k, {})
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        res[k + "_hbm_bytes_per_launch"] = 2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
