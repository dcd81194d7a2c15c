// affinity.hpp -- which CPUs the service threads of the engine on GPU <bdf> run on (host arithmetic over sysfs; no HIP).
//
// at 8 ranks "the >= 6x target is about host feeding (pinned buffers, async H2D, one submit thread per
// GPU)" (server-node/src/services/restorator.js:198-211: every image is an independent job).  Eight ranks on one host each run a
// launcher, a completer and (Node) four waiter threads and move 6 MB per image through pinned memory: left to the scheduler they
// land on any socket.  The plan: a rank's threads stay on the NUMA node its GPU hangs off (/sys/bus/pci/devices/<bdf>/numa_node),
// and the GPUs that share a node split that node's CPUs between them -- every range of the node's cpulist (physical cores |
// their SMT siblings) is cut into as many equal pieces as the node has GPUs and GPU number k (by PCI address) takes piece k of
// each, so sibling threads stay together and two ranks never share a core.  Pinned staging needs no extra step: hipHostMalloc
// without hipHostMallocNumaUser places the pages on the node closest to the current device.
#pragma once
#include <dirent.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

namespace ire {

struct CpuPlan {
    int numa_node = -1;
    int slot = 0, nslots = 1;                       // this GPU's position among the GPUs of its node
    std::vector<std::pair<int, int>> ranges;        // inclusive CPU ranges of the plan; empty: no binding
    std::vector<int> cpus() const {
        std::vector<int> v;
        for (auto& r : ranges) for (int c = r.first; c <= r.second; ++c) v.push_back(c);
        return v;
    }
    std::string cpulist() const {
        std::string s;
        for (auto& r : ranges) {
            if (!s.empty()) s += ",";
            s += std::to_string(r.first);
            if (r.second != r.first) s += "-" + std::to_string(r.second);
        }
        return s;
    }
};

inline bool read_line(const std::string& path, std::string* out) {
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!ok) return false;
    std::string s(buf);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
    *out = s;
    return true;
}

// "0-63,128-191" -> {(0,63),(128,191)}; malformed pieces are skipped
inline std::vector<std::pair<int, int>> parse_cpulist(const std::string& s) {
    std::vector<std::pair<int, int>> v;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string p = s.substr(i, j - i);
        i = j + 1;
        if (p.empty()) continue;
        char* end = nullptr;
        const long a = std::strtol(p.c_str(), &end, 10);
        if (end == p.c_str() || a < 0) continue;
        long b = a;
        if (*end == '-') { char* e2 = nullptr; b = std::strtol(end + 1, &e2, 10); if (e2 == end + 1 || b < a) continue; }
        v.emplace_back((int)a, (int)b);
    }
    return v;
}

// AMD display / accelerator functions (vendor 0x1002; class 0x03xxxx display or 0x12xxxx processing accelerator)
inline bool is_amd_gpu(const std::string& dev_dir) {
    std::string vendor, cls;
    if (!read_line(dev_dir + "/vendor", &vendor) || !read_line(dev_dir + "/class", &cls)) return false;
    if (std::strtol(vendor.c_str(), nullptr, 16) != 0x1002) return false;
    const long c = std::strtol(cls.c_str(), nullptr, 16) >> 16;
    return c == 0x03 || c == 0x12;
}

inline std::string lower(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }

// sysfs_root: "/sys" on a real host (tests pass a fabricated tree).  A missing file or a node of -1 yields an empty plan.
inline CpuPlan affinity_plan(const std::string& sysfs_root, const std::string& bdf_in) {
    CpuPlan p;
    const std::string bdf = lower(bdf_in), devs = sysfs_root + "/bus/pci/devices";
    std::string s;
    if (!read_line(devs + "/" + bdf + "/numa_node", &s)) return p;
    p.numa_node = std::atoi(s.c_str());
    if (p.numa_node < 0) return p;
    if (!read_line(sysfs_root + "/devices/system/node/node" + std::to_string(p.numa_node) + "/cpulist", &s)) return p;
    const auto node = parse_cpulist(s);
    if (node.empty()) return p;
    std::vector<std::string> peers;
    if (DIR* d = opendir(devs.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string name = lower(e->d_name);
            if (name.empty() || name[0] == '.') continue;
            std::string nn;
            if (!read_line(devs + "/" + name + "/numa_node", &nn) || std::atoi(nn.c_str()) != p.numa_node) continue;
            if (name == bdf || is_amd_gpu(devs + "/" + name)) peers.push_back(name);
        }
        closedir(d);
    }
    std::sort(peers.begin(), peers.end());
    peers.erase(std::unique(peers.begin(), peers.end()), peers.end());
    p.nslots = std::max<int>(1, (int)peers.size());
    p.slot = (int)(std::find(peers.begin(), peers.end(), bdf) - peers.begin());
    if (p.slot >= p.nslots) p.slot = 0;
    for (auto& r : node) {
        const int n = r.second - r.first + 1;
        if (n < p.nslots) continue;                 // a range with fewer CPUs than GPUs: left to nobody rather than shared
        const int lo = r.first + (int)((long long)n * p.slot / p.nslots), hi = r.first + (int)((long long)n * (p.slot + 1) / p.nslots) - 1;
        if (hi >= lo) p.ranges.emplace_back(lo, hi);
    }
    if (p.ranges.empty()) p.ranges = node;          // (a node with fewer CPUs than GPUs in every range: the whole node, shared)
    return p;
}

}  // namespace ire
