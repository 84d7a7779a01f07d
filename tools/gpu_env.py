"""What distinguishes one GPU box from another, recorded beside every latency number (VERDICT r3 "Next round" #1):
clocks, power cap, performance level, partition modes, hardware-queue budget, identity.  `python tools/gpu_env.py` prints one
JSON object; bench.py stores `snapshot()` in its detail file as `qc_verify_latency.env`.  (Round 4's probe programs - lone-wavefront
clock, dependent gathers, instruction feed - left the tree in round 5: the library now times the kernel forms themselves at
init, hs_ed25519_autotune, and bench.py reports both forms on the box it runs on.)  Everything is best-effort: a missing tool or field is recorded
as null, never an error."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout=20):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        return r.stdout if r.returncode == 0 else None
    except (OSError, subprocess.TimeoutExpired):
        return None


def rocm_smi():
    """sclk / mclk / fclk / socclk current levels, performance level, power (average, cap), partition modes - card 0."""
    out = {}
    txt = _run(["rocm-smi", "-d", "0", "--showclocks", "--showperflevel", "--showpower", "--showmaxpower", "--showcomputepartition",
                "--showmemorypartition", "--showsclkrange", "--showtemp", "--json"])
    if not txt:
        return None
    try:
        j = json.loads(txt)
    except ValueError:
        return {"raw": txt[:2000]}
    card = j.get("card0") or (list(j.values())[0] if j else {})
    for k, v in card.items():
        kl = k.lower()
        if "sclk clock level" in kl or kl.startswith("sclk"):
            out.setdefault("sclk", v)
        elif "mclk" in kl:
            out.setdefault("mclk", v)
        elif "fclk" in kl:
            out.setdefault("fclk", v)
        elif "socclk" in kl:
            out.setdefault("socclk", v)
        elif "performance level" in kl:
            out["perf_level"] = v
        elif "max graphics package power" in kl or "power cap" in kl:
            out["power_cap_w"] = v
        elif "power" in kl and "average" in kl or "socket graphics package power" in kl:
            out.setdefault("power_w", v)
        elif "compute partition" in kl:
            out["compute_partition"] = v
        elif "memory partition" in kl:
            out["memory_partition"] = v
        elif "valid sclk range" in kl:
            out["sclk_range"] = v
        elif "temperature" in kl and "junction" in kl:
            out["temp_junction_c"] = v
    out["fields_seen"] = len(card)
    return out


def kfd_params():
    """How the kernel driver schedules user queues: with the hardware scheduler (sched_policy 0) a GPU runs at most a fixed
    number of user queues at once; more than that across ALL processes and the runlist is oversubscribed - queues are then
    time-sliced, which is what a fifth idle GPU process (or a few idle stream contexts too many) costs everybody else."""
    out = {}
    for name in ("sched_policy", "hws_max_conc_proc", "num_kcq", "max_num_of_queues_per_device", "queue_preemption_timeout_ms", "mes", "halt_if_hws_hang"):
        try:
            out[name] = open("/sys/module/amdgpu/parameters/" + name).read().strip()
        except OSError:
            out[name] = None
    try:
        import glob
        procs = glob.glob("/sys/class/kfd/kfd/proc/*")
        out["kfd_processes"] = len(procs)
        q = 0
        for p in procs:
            q += len(glob.glob(p + "/queues/*"))
        out["kfd_queues_total"] = q
    except OSError:
        pass
    return out


def identity():
    """WHICH box and which GPU of it: unique id, PCI location, VBIOS, firmware and driver versions, the KFD topology node's
    properties, the NUMA node the card hangs off - whatever could tell the two 'classes of box' apart when clocks do not."""
    out = {"uname": _run(["uname", "-r"]), "hostname": _run(["hostname"])}
    for k in ("uname", "hostname"):
        out[k] = out[k].strip() if out[k] else None
    try:
        out["amdgpu_version"] = open("/sys/module/amdgpu/version").read().strip()
    except OSError:
        out["amdgpu_version"] = None
    txt = _run(["rocm-smi", "-d", "0", "--showuniqueid", "--showvbios", "--showbus", "--showserial", "--showfwinfo", "--showdriverversion", "--json"])
    try:
        j = json.loads(txt) if txt else {}
        card = j.get("card0") or {}
        out["unique_id"] = card.get("Unique ID"); out["vbios"] = card.get("VBIOS version"); out["pci_bus"] = card.get("PCI Bus")
        out["serial"] = card.get("Serial Number")
        out["driver"] = (j.get("system") or {}).get("Driver version")
        out["firmware"] = {k.replace(" firmware version", ""): v for k, v in card.items() if "firmware version" in k}
    except ValueError:
        out["rocm_smi_raw"] = (txt or "")[:1500]
    # the KFD topology node of the (first) GPU
    try:
        import glob
        for node in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*")):
            props = dict(l.split(None, 1) for l in open(node + "/properties").read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) == 0:
                continue
            keep = ("simd_count", "array_count", "num_xcc", "cu_per_simd_array", "max_waves_per_simd", "lds_size_in_kb", "fw_version", "sdma_fw_version",
                    "unique_id", "location_id", "domain", "hive_id", "max_engine_clk_fcompute", "num_sdma_engines", "num_sdma_xgmi_engines", "num_cp_queues",
                    "num_sdma_queues_per_engine", "debug_prop", "capability", "gfx_target_version")
            out["kfd_node"] = {k: props.get(k) for k in keep if k in props}
            banks = []
            for b in sorted(glob.glob(node + "/mem_banks/*/properties")):
                bp = dict(l.split(None, 1) for l in open(b).read().splitlines() if " " in l)
                banks.append({k: bp.get(k) for k in ("heap_type", "size_in_bytes", "width", "mem_clk_max")})
            out["kfd_mem_banks"] = banks
            out["kfd_caches"] = len(glob.glob(node + "/caches/*"))
            break
    except (OSError, ValueError):
        pass
    try:
        if out.get("pci_bus"):
            out["numa_node"] = open("/sys/bus/pci/devices/%s/numa_node" % out["pci_bus"].lower()).read().strip()
            out["pcie_link"] = open("/sys/bus/pci/devices/%s/current_link_speed" % out["pci_bus"].lower()).read().strip()
    except OSError:
        pass
    try:
        out["thp"] = open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip()
    except OSError:
        pass
    return out


def snapshot():
    s = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "HS_ED25519_TENANTS": os.environ.get("HS_ED25519_TENANTS"),
         "HSA_ENABLE_SDMA": os.environ.get("HSA_ENABLE_SDMA"), "host_cores": os.cpu_count()}
    s["rocm_smi"] = rocm_smi()
    s["kfd"] = kfd_params()
    s["identity"] = identity()
    try:
        import torch
        if torch.cuda.is_available():
            p = torch.cuda.get_device_properties(0)
            s["device"] = {"name": p.name, "cus": p.multi_processor_count, "clock_rate_mhz": getattr(p, "clock_rate", 0) // 1000 or None,
                           "hbm_gb": round(p.total_memory / 2**30, 1), "gcn_arch": getattr(p, "gcnArchName", None)}
    except Exception:      # noqa: BLE001
        pass
    return s


if __name__ == "__main__":
    snap = snapshot()
    json.dump(snap, sys.stdout, indent=1)
    print()
