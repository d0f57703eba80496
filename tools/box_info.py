"""What kind of box is this?  Everything about the GPU node that can be READ without privileges and that could
separate the pool's two box classes (identical generic probes, 25 % apart on kernels that start on
their predecessor's output): amdgpu driver / firmware / VBIOS versions, the amdgpu module parameters (mtype_local & co.),
compute / memory partition modes, XNACK, clocks and power cap, kernel command line, NUMA layout, HSA / HIP environment.

    python tools/box_info.py [--full] > gpurun_out/box_info.json

`collect()` returns the full record (every module parameter, every firmware version); `summary()` the dozen fields that
go into every bench line's `box.settings`.  Nothing here touches the device through HIP: sysfs, procfs and the ROCm
command-line tools only; every source that is missing or unreadable is recorded as such instead of raising."""
import glob
import json
import os
import re
import subprocess
import sys


def _read(path, limit=4096):
    try:
        with open(path, "rb") as f:
            return f.read(limit).decode("utf-8", "replace").strip()
    except Exception as e:                                   # unreadable is an answer too
        return "<%s>" % type(e).__name__


def _run(cmd, timeout=20):
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)
        return r.stdout
    except Exception as e:
        return "<%s: %s>" % (type(e).__name__, e)


def _gpu_cards():
    """sysfs device directories of the amdgpu cards that expose compute (vendor 0x1002)."""
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _read(os.path.join(d, "vendor")) == "0x1002":
            out.append(d)
    return out


def collect():
    rec = {}
    rec["uname"] = " ".join(os.uname())
    rec["cmdline"] = _read("/proc/cmdline")
    rec["amdgpu_version"] = _read("/sys/module/amdgpu/version")
    rec["amdgpu_srcversion"] = _read("/sys/module/amdgpu/srcversion")
    params = {}
    for p in sorted(glob.glob("/sys/module/amdgpu/parameters/*")):
        params[os.path.basename(p)] = _read(p, 256)
    rec["amdgpu_parameters"] = params
    cards = []
    for d in _gpu_cards():
        c = {"path": d}
        for name in ("device", "subsystem_device", "revision", "vbios_version", "current_compute_partition",
                     "current_memory_partition", "available_compute_partition", "available_memory_partition",
                     "power_dpm_force_performance_level", "pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk",
                     "mem_info_vram_total", "mem_info_vram_vendor", "numa_node", "local_cpulist", "current_link_speed",
                     "current_link_width", "xgmi_device_id", "xgmi_hive_info/xgmi_hive_id", "unique_id",
                     "pp_features", "gpu_busy_percent", "thermal_throttling_logging"):
            v = _read(os.path.join(d, name), 1024)
            if not v.startswith("<FileNotFound"):
                c[name] = v
        fw = {}
        for p in sorted(glob.glob(os.path.join(d, "fw_version", "*"))):
            fw[os.path.basename(p)] = _read(p, 64)
        c["fw_version"] = fw
        hw = {}
        for p in sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*", "*"))):
            b = os.path.basename(p)
            if re.match(r"(power1_(cap|cap_max|cap_default|average|input)|freq[12]_input|temp[123]_input)$", b):
                hw[b] = _read(p, 64)
        c["hwmon"] = hw
        cards.append(c)
    rec["cards"] = cards
    nodes = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*")):
        props = _read(os.path.join(d, "properties"), 8192)
        kv = dict(l.split(None, 1) for l in props.splitlines() if " " in l)
        if kv.get("simd_count", "0") != "0":
            keep = ("simd_count", "array_count", "simd_arrays_per_engine", "cu_per_simd_array", "max_waves_per_simd",
                    "lds_size_in_kb", "gfx_target_version", "capability", "debug_prop", "sdma_fw_version", "fw_version",
                    "num_xcc", "max_engine_clk_fcompute", "local_mem_size", "num_sdma_engines", "num_sdma_xgmi_engines",
                    "num_cp_queues", "drm_render_minor", "hive_id", "unique_id")
            n = {k: kv[k] for k in keep if k in kv}
            n["node"] = os.path.basename(d)
            mem = []
            for m in sorted(glob.glob(os.path.join(d, "mem_banks", "*", "properties"))):
                mkv = dict(l.split(None, 1) for l in _read(m, 2048).splitlines() if " " in l)
                mem.append({k: mkv[k] for k in ("heap_type", "size_in_bytes", "flags", "width", "mem_clk_max") if k in mkv})
            n["mem_banks"] = mem
            caches = []
            for m in sorted(glob.glob(os.path.join(d, "caches", "*", "properties")))[:64]:
                mkv = dict(l.split(None, 1) for l in _read(m, 2048).splitlines() if " " in l)
                caches.append((mkv.get("level"), mkv.get("size"), mkv.get("type")))
            n["cache_levels"] = sorted({c for c in caches}, key=str)
            nodes.append(n)
    rec["kfd_nodes"] = nodes
    ri = _run(["/opt/rocm/bin/rocminfo"])
    rec["rocminfo_xnack"] = sorted(set(re.findall(r"gfx950[:\w+-]*", ri)))
    rec["rocminfo_agents"] = re.findall(r"Marketing Name:\s+(.*)", ri)
    m = re.search(r"Runtime Version:\s+(\S+)", ri)
    rec["hsa_runtime_version"] = m.group(1) if m else None
    rec["rocminfo_memory_properties"] = sorted(set(x.strip() for x in re.findall(r"Memory Properties:\s*(.*)", ri)))
    rec["rocminfo_coherent_host"] = sorted(set(re.findall(r"Coherent Host Access:\s+(\S+)", ri)))
    for name, cmd in (("amd_smi_static", ["amd-smi", "static", "--json"]),
                      ("amd_smi_metric", ["amd-smi", "metric", "--json"]),
                      ("amd_smi_version", ["amd-smi", "version"]),
                      ("rocm_smi", ["/opt/rocm/bin/rocm-smi", "--showclocks", "--showperflevel", "--showmemvendor",
                                    "--showmaxpower", "--showpower", "--showvbios", "--showdriverversion",
                                    "--showcomputepartition", "--showmemorypartition", "--json"])):
        out = _run(cmd)
        try:
            rec[name] = json.loads(out[out.index("{") if "{" in out and not out.lstrip().startswith("[") else 0:])
        except Exception:
            rec[name] = out[-3000:]
    rec["env"] = {k: v for k, v in os.environ.items()
                  if re.match(r"(HSA_|HIP_|ROCR_|GPU_|AMD_|NCCL_|RCCL_|OMP_|PYTORCH_|MIOPEN_|ROCM_)", k)}
    rec["cpu_model"] = next((l.split(":", 1)[1].strip() for l in _read("/proc/cpuinfo", 1 << 16).splitlines()
                             if l.startswith("model name")), None)
    rec["host_cpus"] = os.cpu_count()
    rec["numa_nodes"] = len(glob.glob("/sys/devices/system/node/node[0-9]*"))
    rec["thp"] = _read("/sys/kernel/mm/transparent_hugepage/enabled")
    rec["iommu_groups"] = len(glob.glob("/sys/kernel/iommu_groups/*"))
    rec["meminfo_total_kB"] = next((l.split()[1] for l in _read("/proc/meminfo").splitlines() if l.startswith("MemTotal")), None)
    return rec


_PARAMS_OF_INTEREST = ("mtype_local", "noretry", "vm_fragment_size", "vm_update_mode", "sched_policy", "hws_max_conc_proc",
                       "cwsr_enable", "mes", "mes_kiq", "ras_enable", "tmz", "gpu_recovery", "pcie_gen_cap", "aspm",
                       "runpm", "ppfeaturemask", "mcbp", "debug_evictions", "no_queue_eviction_on_vm_fault", "svm_default_granularity",
                       "use_xgmi_p2p", "user_partt_mode", "send_sigterm", "halt_if_hws_hang", "queue_preemption_timeout_ms")


def active_card(rec):
    """The card THIS process computes on: the node's sysfs lists every GPU of the (shared) host, the KFD topology only the
    one(s) the lease exposes — matched by unique_id (KFD prints it in decimal, the card in hex)."""
    ids = set()
    for n in rec.get("kfd_nodes", []):
        try:
            ids.add("%016x" % int(n.get("unique_id", "")))
        except ValueError:
            pass
    for c in rec.get("cards", []):
        if c.get("unique_id") in ids:
            return c
    return rec["cards"][0] if rec.get("cards") else {}


def neighbours(rec):
    """[busy %, watts] of the OTHER cards of the host at the moment of the reading (other tenants' jobs)."""
    mine = active_card(rec).get("path")
    out = []
    for c in rec.get("cards", []):
        if c.get("path") != mine:
            try:
                out.append([int(c.get("gpu_busy_percent", "-1")), int(c.get("hwmon", {}).get("power1_input", "0")) // 1000000])
            except ValueError:
                out.append([None, None])
    return out


def sample_active(seconds=2.0, hz=20):
    """clock / power / busy readings of the active card and the neighbours' load while something else runs on it (called
    from a thread next to the timed region): min / median / max over the samples."""
    import time
    rec = {"cards": [], "kfd_nodes": collect_kfd_ids()}
    paths = _gpu_cards()
    cards = [{"path": d, "unique_id": _read(os.path.join(d, "unique_id"), 64)} for d in paths]
    rec["cards"] = cards
    mine = active_card(rec).get("path") or (paths[0] if paths else None)
    if mine is None:
        return {}
    hw = sorted(glob.glob(os.path.join(mine, "hwmon", "hwmon*")))
    hw = hw[0] if hw else None
    series = {"sclk_MHz": [], "mclk_MHz": [], "power_W": [], "busy": [], "temp_hot_C": [], "neighbours_busy": [], "neighbours_W": []}
    t_end = time.time() + seconds
    while time.time() < t_end:
        if hw:
            for key, f, div in (("sclk_MHz", "freq1_input", 1e6), ("mclk_MHz", "freq2_input", 1e6), ("power_W", "power1_input", 1e6),
                                ("temp_hot_C", "temp2_input", 1e3)):
                try:
                    series[key].append(float(_read(os.path.join(hw, f), 32)) / div)
                except ValueError:
                    pass
        try:
            series["busy"].append(float(_read(os.path.join(mine, "gpu_busy_percent"), 16)))
        except ValueError:
            pass
        nb, nw = 0.0, 0.0
        for d in paths:
            if d != mine:
                try:
                    nb += float(_read(os.path.join(d, "gpu_busy_percent"), 16))
                    h = sorted(glob.glob(os.path.join(d, "hwmon", "hwmon*")))
                    nw += float(_read(os.path.join(h[0], "power1_input"), 32)) / 1e6 if h else 0.0
                except ValueError:
                    pass
        series["neighbours_busy"].append(nb)
        series["neighbours_W"].append(nw)
        time.sleep(1.0 / hz)
    out = {"card": os.path.basename(os.path.dirname(mine)), "samples": len(series["busy"])}
    for k, v in series.items():
        if v:
            v = sorted(v)
            out[k] = [round(v[0], 1), round(v[len(v) // 2], 1), round(v[-1], 1)]
    return out


def collect_kfd_ids():
    out = []
    for d in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*")):
        kv = dict(l.split(None, 1) for l in _read(os.path.join(d, "properties"), 8192).splitlines() if " " in l)
        if kv.get("simd_count", "0") != "0":
            out.append({"unique_id": kv.get("unique_id", "")})
    return out


def summary(rec=None):
    """the fields that go into every bench line (`box.settings`): short strings only."""
    rec = rec or collect()
    c0 = active_card(rec)
    n0 = rec["kfd_nodes"][0] if rec.get("kfd_nodes") else {}
    def cur(s):                               # the starred line of a pp_dpm_* table
        if not isinstance(s, str):
            return None
        lines = [l for l in s.splitlines() if l.rstrip().endswith("*")]
        return lines[0].split(":", 1)[1].strip(" *") if lines else (s.splitlines()[-1] if s else None)
    fw = c0.get("fw_version", {})
    out = {
        "kernel": rec.get("uname", "").split(" ")[2] if rec.get("uname") else None,
        "amdgpu_version": rec.get("amdgpu_version"),
        "vbios": c0.get("vbios_version"),
        "fw": {k: fw.get(k) for k in ("mec_fw_version", "mec2_fw_version", "rlc_fw_version", "smc_fw_version",
                                      "sdma_fw_version", "psp_sos_fw_version", "mes_fw_version", "imu_fw_version")
               if fw.get(k) and not str(fw.get(k)).startswith("<")},
        "amdgpu_params": {k: rec.get("amdgpu_parameters", {}).get(k) for k in _PARAMS_OF_INTEREST
                          if k in rec.get("amdgpu_parameters", {})},
        "compute_partition": c0.get("current_compute_partition"),
        "memory_partition": c0.get("current_memory_partition"),
        "perf_level": c0.get("power_dpm_force_performance_level"),
        "sclk": cur(c0.get("pp_dpm_sclk")), "mclk": cur(c0.get("pp_dpm_mclk")), "fclk": cur(c0.get("pp_dpm_fclk")),
        "socclk": cur(c0.get("pp_dpm_socclk")),
        "power_cap_uW": c0.get("hwmon", {}).get("power1_cap"),
        "gpu_numa_node": c0.get("numa_node"),
        "card": os.path.basename(os.path.dirname(c0.get("path", "/x/x"))), "gpu_unique_id": c0.get("unique_id"),
        "pcie": "%s x%s" % (c0.get("current_link_speed"), c0.get("current_link_width")),
        "vram_vendor": c0.get("mem_info_vram_vendor"),
        "n_cards": len(rec.get("cards", [])),
        "neighbours_busy_W": neighbours(rec),
        "xnack": rec.get("rocminfo_xnack"),
        "kfd_capability": n0.get("capability"), "kfd_debug_prop": n0.get("debug_prop"), "num_xcc": n0.get("num_xcc"),
        "mem_width_clk": [(m.get("width"), m.get("mem_clk_max")) for m in n0.get("mem_banks", [])][:1],
        "cpu": rec.get("cpu_model"), "host_cpus": rec.get("host_cpus"), "numa_nodes": rec.get("numa_nodes"),
        "cmdline_flags": [w for w in (rec.get("cmdline") or "").split()
                          if re.match(r"(iommu|amd_iommu|amdgpu\.|pci=|numa_balancing|transparent_hugepage|mitigations|pcie_aspm|processor\.|idle=|intel_idle|nohz|isolcpus)", w)],
        "thp": rec.get("thp"),
        "env": rec.get("env"),
    }
    return out


if __name__ == "__main__":
    if "--sample" in sys.argv:             # bench.py runs this next to its timed region (a process of its own: no GIL share)
        print(json.dumps(sample_active(float(sys.argv[sys.argv.index("--sample") + 1]))))
        sys.exit(0)
    r = collect()
    if "--full" in sys.argv:
        print(json.dumps(r, indent=1, default=str))
    else:
        print(json.dumps({"summary": summary(r), "full": r}, indent=1, default=str))
