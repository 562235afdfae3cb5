// oracle_model.hpp — TEST INFRASTRUCTURE.  CPU restatement of the reference's session data model.
//
// This directory is the parity oracle.  Nothing in the product library links,
// includes or executes it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
// and only as the checker.  It restates the reference algorithm in plain single-threaded C++ with
// the reference's own evaluation order; every type cites the Go file it follows
// (paths relative to /root/reference/pkg/scheduler).
//
// Pinning: the restatement is accepted against the reference's own golden tables transcribed by
// tools/go_fixtures.py into tests/golden/*.json (see tests/test_oracle_golden.py).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../include/kai_core.h"

namespace orc {

// ---------------------------------------------------------------- pod status: api/pod_status/pod_status.go:25-71
enum : int {
    Pending = KAI_POD_PENDING, Gated = KAI_POD_GATED, Allocated = KAI_POD_ALLOCATED, Pipelined = KAI_POD_PIPELINED,
    Binding = KAI_POD_BINDING, Bound = KAI_POD_BOUND, Running = KAI_POD_RUNNING, Releasing = KAI_POD_RELEASING,
    Succeeded = KAI_POD_SUCCEEDED, Failed = KAI_POD_FAILED, Unknown = KAI_POD_UNKNOWN, Deleted = KAI_POD_DELETED
};
inline bool IsActiveUsedStatus(int s) { return s & (Allocated | Pipelined | Binding | Bound | Running | Releasing); }
inline bool IsActiveAllocatedStatus(int s) { return s & (Allocated | Pipelined | Binding | Bound | Running); }
inline bool IsAliveStatus(int s) { return s & (Allocated | Pipelined | Binding | Bound | Running | Pending | Gated); }
inline bool AllocatedStatus(int s) { return s & (Allocated | Bound | Binding | Running); }

// ---------------------------------------------------------------- api/resource_info/base_resources.go
// scalar resources are keyed by the resource-vector index (3 = pods, 4.. = extras)
struct BaseResource {
    double milliCpu = 0, memory = 0;
    std::map<int, int64_t> scalars;
    void Add(const BaseResource& o) {  // base_resources.go:56-63 (deletes a key that reaches 0)
        milliCpu += o.milliCpu; memory += o.memory;
        for (auto& kv : o.scalars) { scalars[kv.first] += kv.second; if (scalars[kv.first] == 0) scalars.erase(kv.first); }
    }
    void Sub(const BaseResource& o) {  // base_resources.go:65-72
        milliCpu -= o.milliCpu; memory -= o.memory;
        for (auto& kv : o.scalars) { scalars[kv.first] -= kv.second; if (scalars[kv.first] == 0) scalars.erase(kv.first); }
    }
    double GetScalar(int k) const { auto it = scalars.find(k); return it == scalars.end() ? 0.0 : double(it->second); }
    bool LessEqual(const BaseResource& rr) const {  // base_resources.go:90-105
        if (milliCpu > rr.milliCpu) return false;
        if (memory > rr.memory) return false;
        for (auto& kv : scalars) { auto it = rr.scalars.find(kv.first); if (it == rr.scalars.end() || kv.second > it->second) return false; }
        return true;
    }
    bool IsEmpty() const {  // base_resources.go:119-130
        if (milliCpu >= 10.0 || memory >= 10.0 * 1024 * 1024) return false;
        for (auto& kv : scalars) if (kv.second >= 10) return false;
        return true;
    }
};

// MIG profiles (api/common_info/resources/mig.go:13-33): a resource row >= 4 that stands for nvidia.com/mig-<g>g.<m>gb carries its GPU weight g and its memory m
struct MigRows { int gpus[KAI_MAX_RES] = {0}; int64_t mem[KAI_MAX_RES] = {0}; bool any = false; };
inline MigRows& migRows() { static MigRows m; return m; }
inline double migGpus(const std::map<int, int64_t>& scalars) { double q = 0; if (migRows().any) for (auto& kv : scalars) if (kv.first < KAI_MAX_RES && migRows().gpus[kv.first] > 0) q += double(migRows().gpus[kv.first]) * double(kv.second); return q; }

// api/resource_info/resource_info.go:16-19
struct Resource : BaseResource {
    double gpus = 0;
    double GetTotalGPURequest() const { return migGpus(scalars) + gpus; }  // resource_info.go:177-194
    void Add(const Resource& o) { BaseResource::Add(o); gpus += o.gpus; }
    void Sub(const Resource& o) { BaseResource::Sub(o); gpus -= o.gpus; }
    double Get(int r) const { return r == KAI_RES_CPU ? milliCpu : r == KAI_RES_MEM ? memory : r == KAI_RES_GPU ? gpus : GetScalar(r); }
};

// api/resource_info/gpu_resource_requirment.go:230-234 — fixed-point GPU amount
inline double getExtendedResourceGpus(double portion, int64_t count) {
    int64_t portionAsDecimals = (int64_t)std::llround(portion * 100.0);  // math.Round: half away from zero
    return double(portionAsDecimals * count) / 100.0;
}

// api/resource_info/resource_requirment.go:17-20 (+ gpu_resource_requirment.go:26-32; whole GPUs, a fraction of one device, MiB of one device:
// MIG / DRA pods are flagged KAI_POD_CPU_FALLBACK and never reach the path)
struct ResourceRequirements : BaseResource {
    int64_t count = 0; double portion = 0;
    int64_t gpuMemory = 0;  // a request for that many MiB of one device (annotation gpu-memory, pod_info.go:463-468): count 1, portion 0
    double GPUs() const { return getExtendedResourceGpus(portion, count); }
    // MIG instances: the reference keeps them in GpuResourceRequirement.migResources; here they are the scalar rows the MIG table names
    bool HasMig() const { if (migRows().any) for (auto& kv : scalars) if (kv.first < KAI_MAX_RES && migRows().gpus[kv.first] > 0 && kv.second > 0) return true; return false; }
    double GetGpusQuota() const { return migGpus(scalars) + GPUs(); }  // gpu_resource_requirment.go:163-178 (no DRA claims on the path)
    bool IsEmpty() const {                          // resource_requirment.go:99-104, gpu_resource_requirment.go:89-104
        if (GPUs() > 0.01) return false;
        if (HasMig()) return false;
        if (milliCpu >= 10.0 || memory >= 10.0 * 1024 * 1024) return false;  // base_resources.go:119-130 over the scalars that are not MIG instances
        for (auto& kv : scalars) { if (migRows().any && kv.first < KAI_MAX_RES && migRows().gpus[kv.first] > 0) continue; if (kv.second >= 10) return false; }
        return true;
    }
    bool LessEqualResource(const Resource& rr) const {  // resource_requirment.go:126-140
        if (!BaseResource::LessEqual(rr)) return false;
        if (GPUs() > rr.gpus) return false;
        return true;
    }
    Resource AsResource() const {  // what Resource.AddResourceRequirements adds (resource_info.go:143-155)
        Resource r; r.milliCpu = milliCpu; r.memory = memory; r.scalars = scalars; r.gpus = GPUs(); return r;
    }
};

struct PodGroupInfo; struct PodSet; struct NodeInfo;

// api/topology_info: {Topology, RequiredLevel, PreferredLevel}; topology -1 = no constraint, -2 = the named topology does not exist
struct TopologyConstraint { int topology = -1, required = -1, preferred = -1; };

// ---------------------------------------------------------------- api/pod_info/pod_info.go:70-112
struct PodInfo {
    int idx = -1;            // index in the snapshot (stands for UID; ordering uses uidRank)
    uint32_t uidRank = 0;
    int job = -1, podset = -1;
    int status = Pending;
    int node = -1;           // NodeName
    bool isVirtualStatus = false;
    uint32_t flags = 0;
    int taskPriority = 0;    // task-order label
    int64_t createdNs = 0;
    int podClass = 0;
    int nominatedNode = -1;
    ResourceRequirements resReq;
    ResourceRequirements accepted;  // AcceptedResource, set by NodeInfo.setAcceptedResources
    // shared GPUs (ABI v4): a fraction of one device (pod_info.go:472-477 RequestTypeFraction); the groups it holds (PodInfo.GPUGroups; ids are
    // unique per node, ids >= kNewGpuGroup were created in this session — the reference draws a UUID there, gpu_sharing/gpuSharing.go:94)
    bool isFractionRequest = false;   // ResourceRequestType == RequestTypeFraction
    bool receivedFraction = false;    // ResourceReceivedType == ReceivedTypeFraction (node_info.go:755-758)
    std::vector<int> gpuGroups;
    bool isMemoryRequest = false;     // ResourceRequestType == RequestTypeGpuMemory (pod_info.go:463-468)
    bool isMigRequest = false;        // RequestTypeMigInstance: the request holds MIG instances (pod_info.go:493-497)
    bool isLegacyMig = false;         // IsLegacyMIGtask (pod_info.go:500-516)
    bool IsMigCandidate() const { return isMigRequest; }        // pod_info.go:320-322
    bool IsMigProfileRequest() const { return isMigRequest; }   // :300-302
    bool IsMemoryRequest() const { return isMemoryRequest; }                                  // pod_info.go:324-326
    bool IsFractionCandidate() const { return isFractionRequest || isMemoryRequest; }        // :316-318
    bool IsSharedGPURequest() const { return isFractionRequest || isMemoryRequest; }         // :332-334
    bool IsSharedGPUAllocation() const { return receivedFraction; }    // :336-338
    bool IsRegularGPURequest() const { return !isFractionRequest && !isMemoryRequest && !isMigRequest; }     // :328-330 (RequestTypeRegular)
    bool IsCPUOnlyRequest() const { return !(resReq.GPUs() > 0 || isMemoryRequest || isMigRequest); }  // pod_info.go:340-347 IsRequireAnyKindOfGPU
    bool ShouldAllocate(bool isRealAllocation) const {               // pod_info.go:518-521
        return status == Pending || (!isRealAllocation && status == Releasing && isVirtualStatus);
    }
};

// ---------------------------------------------------------------- api/podgroup_info/subgroup_info/podset.go
struct PodSet {
    int idx = -1; uint32_t nameRank = 0; int job = -1;
    int group = -1; TopologyConstraint tc;  // parent SubGroupSet, own constraint (subgroup_info.go)
    int32_t minAvailable = 1;
    std::map<int, PodInfo*> podInfos;  // keyed by pod idx (UID)
    std::map<int, int> podStatusMap;
    int numActiveAllocatedTasks = 0, numActiveUsedTasks = 0, numAliveTasks = 0, numGated = 0;
    void clearOldStatus(PodInfo* ti) {  // podset.go:82-99
        auto it = podStatusMap.find(ti->idx); if (it == podStatusMap.end()) return;
        int old = it->second;
        if (IsActiveAllocatedStatus(old)) numActiveAllocatedTasks -= 1;
        if (IsActiveUsedStatus(old)) numActiveUsedTasks -= 1;
        if (IsAliveStatus(old)) numAliveTasks -= 1;
        if (old == Gated) numGated -= 1;
        podStatusMap.erase(it); podInfos.erase(ti->idx);
    }
    void AssignTask(PodInfo* ti) {  // podset.go:56-77
        clearOldStatus(ti);
        if (IsActiveAllocatedStatus(ti->status)) numActiveAllocatedTasks += 1;
        if (IsActiveUsedStatus(ti->status)) numActiveUsedTasks += 1;
        if (IsAliveStatus(ti->status)) numAliveTasks += 1;
        if (ti->status == Gated) numGated += 1;
        podStatusMap[ti->idx] = ti->status; podInfos[ti->idx] = ti;
    }
    bool IsReadyForScheduling() const { return int32_t(numAliveTasks - numGated) >= minAvailable; }  // podset.go:114-120
    bool IsGangSatisfied() const { return numActiveUsedTasks >= int(minAvailable); }                 // podset.go:122-125
    bool IsElastic() const { return minAvailable < int32_t(podInfos.size()); }
    int GetNumPendingTasks() const { int n = 0; for (auto& kv : podStatusMap) if (kv.second == Pending) n++; return n; }  // podset.go:147-149 (len of the status index's Pending entry)
};

// ---------------------------------------------------------------- api/podgroup_info/subgroup_info/subgroupset.go
struct SubGroupSet {
    int idx = -1, job = -1, parent = -1; uint32_t nameRank = 0; TopologyConstraint tc;
    std::vector<int> groups;   // child SubGroupSets (insertion order)
    std::vector<int> podSets;  // child pod-sets (insertion order)
};

// ---------------------------------------------------------------- plugins/topology/topology_structs.go:41-59
struct DomainInfo {
    int id = -1, topo = -1, level = -1 /* inside the topology, 0 = top; -1 = the root domain */, parent = -1; uint32_t idRank = 0;
    std::vector<int> children;  // ordered: sortTree re-orders them in place
    std::vector<int> nodes;
    int AllocatablePods = -1;   // allocatablePodsNotSet
    Resource IdleOrReleasingResources;
};

// ---------------------------------------------------------------- api/podgroup_info/job_info.go:65-103
struct PodGroupInfo {
    int idx = -1; uint32_t uidRank = 0;
    int queue = -1; int32_t priority = 0; bool preemptible = true; int64_t createdNs = 0;
    int rootGroup = -1;            // RootSubGroupSet
    std::vector<PodSet*> podSets;  // GetSubGroups(); kept sorted by name rank (Go ranges a map; order-free uses only)
    Resource allocated;            // job_info.go:78 Allocated
    std::map<int, std::map<int, PodInfo*>> podStatusIndex;
    // inner cache (allocation_info.go:31-33,97-99): NOT keyed by isRealAllocation, exactly like the reference
    bool hasTasksToAllocate = false; std::vector<PodInfo*> tasksToAllocate;
    bool hasInitResource = false; Resource tasksToAllocateInitResource;
    bool hasLastStart = false;
    bool isClone = false;          // CloneWithTasks representative (job_info.go:477-510): same UID, own pod-sets over a subset of the tasks
    int64_t lastStartNs = 0;       // LastStartTimestamp (0 = nil / zero)
    int64_t signature = 0;         // GetSchedulingConstraintsSignature (job_info.go:547-570): any injective id of the constraint set
    std::vector<PodInfo*> AllPods() const { std::vector<PodInfo*> v; for (auto* ps : podSets) for (auto& kv : ps->podInfos) v.push_back(kv.second); return v; }
    std::vector<PodInfo*> AllPodsByIndex() const { std::map<int, PodInfo*> m; for (auto* ps : podSets) for (auto& kv : ps->podInfos) m[kv.first] = kv.second; std::vector<PodInfo*> v; for (auto& kv : m) v.push_back(kv.second); return v; }
    bool IsPreemptibleJob() const { return preemptible; }
    void invalidateTasksCache() { hasTasksToAllocate = false; tasksToAllocate.clear(); hasInitResource = false; }
    PodSet* podSetByIdx(int k) const { for (auto* ps : podSets) if (ps->idx == k) return ps; return nullptr; }
    PodSet* podSetOf(PodInfo* t) const { for (auto* ps : podSets) if (ps->idx == t->podset) return ps; return nullptr; }
    void AddTaskInfo(PodInfo* ti) {  // job_info.go:208-226
        PodSet* ps = podSetOf(ti); if (!ps) return;
        ps->AssignTask(ti);
        podStatusIndex[ti->status][ti->idx] = ti; invalidateTasksCache();
        if (AllocatedStatus(ti->status)) allocated.Add(ti->resReq.AsResource());
    }
    void UpdateTaskStatus(PodInfo* task, int status) {  // job_info.go:228-238 + resetTaskState :272-287
        if (AllocatedStatus(task->status)) allocated.Sub(task->resReq.AsResource());
        auto it = podStatusIndex.find(task->status);
        if (it != podStatusIndex.end()) { it->second.erase(task->idx); if (it->second.empty()) podStatusIndex.erase(it); invalidateTasksCache(); }
        task->status = status;
        AddTaskInfo(task);
    }
    int GetNumAllocatedTasks() const { int n = 0; for (auto* t : AllPods()) if (AllocatedStatus(t->status)) n++; return n; }
    int GetNumPendingTasks() const { auto it = podStatusIndex.find(Pending); return it == podStatusIndex.end() ? 0 : int(it->second.size()); }
    bool IsReadyForScheduling() const { for (auto* ps : podSets) if (!ps->IsReadyForScheduling()) return false; return true; }  // job_info.go:399-406
    bool IsGangSatisfied() const { for (auto* ps : podSets) if (!ps->IsGangSatisfied()) return false; return true; }
    bool ShouldPipelineJob() const {  // job_info.go:443-464
        for (auto* ps : podSets) {
            bool hasPipelinedTask = false; int activeAllocated = 0;
            for (auto& kv : ps->podInfos) {
                if (kv.second->status == Pipelined) hasPipelinedTask = true;
                else if (IsActiveAllocatedStatus(kv.second->status)) activeAllocated += 1;
            }
            if (hasPipelinedTask && activeAllocated < int(ps->minAvailable)) return true;
        }
        return false;
    }
};

// ---------------------------------------------------------------- api/node_info/node_info.go:68-105
constexpr int kNewGpuGroup = 1 << 20;  // group ids from here on were created by this session (a UUID in the reference); below: groups of the snapshot ("0", "1", …)
constexpr int kWholeGpuIndicator = -1; // pod_info.WholeGpuIndicator

struct NodeInfo {
    int idx = -1; uint32_t nameRank = 0; uint32_t flags = 0; int gpuCountLabel = -1; int nodeClass = 0;
    Resource Idle, Used, Releasing, Allocatable;
    int64_t MemoryOfEveryGpuOnNode = 100;  // node_info.go:48 DefaultGpuMemory
    // GpuSharingNodeInfo (api/node_info/gpu_sharing_node_info.go:17-27), keyed by group id
    std::set<int> ReleasingSharedGPUs;
    std::map<int, int64_t> UsedSharedGPUsMemory, ReleasingSharedGPUsMemory, AllocatedSharedGPUsMemory;
    struct OnNode { int status; Resource tracked; bool shared; std::vector<int> groups; int64_t gpuMemory; };
    std::map<int, OnNode> podInfos;  // the node's own copy of the task (status, groups and amounts at add time)
    double NonAllocatedResource(int r) const { return Idle.Get(r) + Releasing.Get(r); }  // node_info.go:164-166
    Resource NonAllocatedResources() const { Resource x; x.Add(Idle); x.Add(Releasing); return x; }  // :157-162
    bool IsCPUOnlyNode() const { if (flags & KAI_NODE_MIG_ENABLED) return false; return Allocatable.gpus <= 0 && !(flags & KAI_NODE_HAS_DRA_GPUS); }  // :697-702
    int64_t GetNumberOfGPUsInNode() const { return gpuCountLabel >= 0 ? gpuCountLabel : int64_t(Allocatable.gpus); }  // :630-637
    // ---- shared-GPU helpers (gpu_sharing_node_info.go)
    static int64_t get(const std::map<int, int64_t>& m, int g) { auto it = m.find(g); return it == m.end() ? 0 : it->second; }
    int64_t GetResourceGpuMemory(const ResourceRequirements& r) const { return r.gpuMemory > 0 ? r.gpuMemory : int64_t(r.portion * double(MemoryOfEveryGpuOnNode)); }  // node_info.go:653-659
    double getResourceGpuPortion(const ResourceRequirements& r) const { return r.gpuMemory > 0 ? getGpuMemoryFractionalOnNode(r.gpuMemory) : r.portion; }  // :661-666
    bool isValidGpuPortion(const ResourceRequirements& r) const { double p = getResourceGpuPortion(r); return p <= 1 || p == double(int(p)); }  // :668-671
    double getGpuMemoryFractionalOnNode(int64_t memory) const { return std::ceil(double(memory) / double(MemoryOfEveryGpuOnNode) * 100) / 100; }  // :329-332
    int getNumberOfUsedSharedGPUs() const { int n = 0; for (auto& kv : UsedSharedGPUsMemory) if (kv.second > 0) n++; return n; }  // :265-273
    int getNumberOfUsedGPUs() const { return int(Used.gpus) + getNumberOfUsedSharedGPUs(); }                                     // :275-277
    bool isSharedGpuMarkedAsReleasing(int g) const { return ReleasingSharedGPUs.count(g) != 0; }
    bool isGpuReleasingFromSharedTasks(int g) const {  // :253-263
        int64_t used = get(UsedSharedGPUsMemory, g); if (used == 0) return false;
        return ReleasingSharedGPUsMemory.count(g) && get(ReleasingSharedGPUsMemory, g) == used;
    }
    bool enoughResourcesOnGpu(const ResourceRequirements& r, int g) const {  // :365-371
        return MemoryOfEveryGpuOnNode - get(AllocatedSharedGPUsMemory, g) + get(ReleasingSharedGPUsMemory, g) - GetResourceGpuMemory(r) >= 0;
    }
    bool isAllGpuReleased(int g) const { return get(AllocatedSharedGPUsMemory, g) == get(ReleasingSharedGPUsMemory, g); }  // :373-375
    bool IsTaskFitOnGpuGroup(const ResourceRequirements& r, int g) const {  // :350-354
        return get(UsedSharedGPUsMemory, g) != 0 && enoughResourcesOnGpu(r, g) && !isAllGpuReleased(g);
    }
    bool EnoughIdleResourcesOnGpu(const ResourceRequirements& r, int g) const {  // :356-363
        if (!AllocatedSharedGPUsMemory.count(g)) return false;
        return MemoryOfEveryGpuOnNode - get(AllocatedSharedGPUsMemory, g) - GetResourceGpuMemory(r) >= 0;
    }
    double GetUsedGpuPortion(int g) const { return double(get(UsedSharedGPUsMemory, g)) / double(MemoryOfEveryGpuOnNode); }  // :377-383
    double getSumOfAvailableSharedGPUs() const {  // :289-300: the free portion of every shared GPU that has something allocated
        double sum = 0; for (auto& kv : AllocatedSharedGPUsMemory) if (kv.second > 0) sum += 1 - getGpuMemoryFractionalOnNode(kv.second); return sum;
    }
    double getSumOfReleasingSharedGPUs() const {  // :302-313: portions being released on shared GPUs that are not released as a whole
        double sum = 0; for (auto& kv : ReleasingSharedGPUsMemory) if (kv.second > 0 && !isGpuReleasingFromSharedTasks(kv.first)) sum += getGpuMemoryFractionalOnNode(kv.second); return sum;
    }
    bool hasLegacyMigTasks = false;  // len(LegacyMIGTasks) > 0: entries are added with the task and never removed (node_info.go:407-409)
    double GetSumOfIdleGPUs() const { return getSumOfAvailableSharedGPUs() + Idle.gpus + migGpus(Idle.scalars); }            // node_info.go:592-609
    int64_t GetSumOfIdleGPUsMemory() const {  // the second result of :592-609 and gpu_sharing_node_info.go:314-326
        int64_t m = 0; for (auto& kv : AllocatedSharedGPUsMemory) if (kv.second > 0) m += MemoryOfEveryGpuOnNode - kv.second;
        return m + int64_t(Idle.gpus + migGpus(Idle.scalars)) * MemoryOfEveryGpuOnNode;
    }
    int64_t GetSumOfReleasingGPUsMemory() const {  // :611-628, :328-339
        int64_t m = 0; for (auto& kv : ReleasingSharedGPUsMemory) if (kv.second > 0 && !isGpuReleasingFromSharedTasks(kv.first)) m += kv.second;
        return m + int64_t(Releasing.gpus + migGpus(Releasing.scalars)) * MemoryOfEveryGpuOnNode;
    }
    double GetSumOfReleasingGPUs() const { return getSumOfReleasingSharedGPUs() + Releasing.gpus + migGpus(Releasing.scalars); }  // :611-628
    int64_t fractionTaskGpusAllocatableDeviceCount(const PodInfo* pod) const {  // :334-348
        int64_t n = 0;
        for (auto& kv : UsedSharedGPUsMemory) if (IsTaskFitOnGpuGroup(pod->resReq, kv.first)) { n++; if (n >= pod->resReq.count) return n; }
        return n;
    }
    bool isTaskAllocatableOnNonAllocatedResources(const PodInfo* task, const Resource& avail) const {  // :361-382
        if (task->IsRegularGPURequest() || task->IsMigProfileRequest()) return task->resReq.LessEqualResource(avail);
        if (!static_cast<const BaseResource&>(task->resReq).LessEqual(avail)) return false;
        if (!isValidGpuPortion(task->resReq)) return false;
        int64_t wholeGpus = int64_t(std::floor(avail.gpus));
        return wholeGpus + fractionTaskGpusAllocatableDeviceCount(task) >= task->resReq.count;
    }
    bool IsTaskAllocatable(const PodInfo* task) const {  // :168-188 (no storage claims on the path)
        if (task->resReq.IsEmpty() && !task->IsMemoryRequest()) return true;
        return isTaskAllocatableOnNonAllocatedResources(task, Idle);
    }
    bool IsTaskAllocatableOnReleasingOrIdle(const PodInfo* task) const {  // :190-206
        return isTaskAllocatableOnNonAllocatedResources(task, NonAllocatedResources());
    }
    void setAcceptedResources(PodInfo* pi) const {  // :746-766
        if (!IsActiveUsedStatus(pi->status)) return;
        pi->accepted = pi->resReq;
        pi->receivedFraction = pi->IsFractionCandidate();  // ReceivedTypeFraction for a fraction candidate, ReceivedTypeRegular otherwise
        if (pi->IsFractionCandidate()) {  // NewGpuResourceRequirementWithMultiFraction(devices, this node's portion, this node's memory) :756-759
            pi->accepted.portion = getResourceGpuPortion(pi->resReq); pi->accepted.gpuMemory = GetResourceGpuMemory(pi->resReq);
        }
    }
    void addTaskResources(const Resource& r, int status) {  // :457-493
        Used.Add(r);
        switch (status) {
            case KAI_POD_RELEASING: Releasing.Add(r); Idle.Sub(r); break;
            case KAI_POD_PIPELINED: Releasing.Sub(r); break;
            default: Idle.Sub(r);
        }
    }
    void removeTaskResources(const Resource& r, int status) {  // :515-551
        Used.Sub(r);
        switch (status) {
            case KAI_POD_RELEASING: Releasing.Sub(r); Idle.Add(r); break;
            case KAI_POD_PIPELINED: Releasing.Add(r); break;
            default: Idle.Add(r);
        }
    }
    void addSharedTaskResourcesPerPodGroup(int status, int64_t mem, int g) {  // gpu_sharing_node_info.go:83-136
        UsedSharedGPUsMemory[g] += mem;
        switch (status) {
            case KAI_POD_RELEASING:
                ReleasingSharedGPUsMemory[g] += mem; AllocatedSharedGPUsMemory[g] += mem;
                if (UsedSharedGPUsMemory[g] == ReleasingSharedGPUsMemory[g]) {
                    if (!isSharedGpuMarkedAsReleasing(g)) { Releasing.gpus += 1; ReleasingSharedGPUs.insert(g); }
                    if (int(GetNumberOfGPUsInNode()) < int(Idle.gpus) + getNumberOfUsedGPUs()) Idle.gpus -= 1;
                }
                break;
            case KAI_POD_PIPELINED:
                ReleasingSharedGPUsMemory[g] -= mem;
                if (UsedSharedGPUsMemory[g] - mem == ReleasingSharedGPUsMemory[g] + mem) Releasing.gpus -= 1;
                break;
            default:
                AllocatedSharedGPUsMemory[g] += mem;
                if (UsedSharedGPUsMemory[g] <= mem) { if (int(GetNumberOfGPUsInNode()) < int(Idle.gpus) + getNumberOfUsedGPUs()) Idle.gpus -= 1; }
                if (isSharedGpuMarkedAsReleasing(g)) { Releasing.gpus -= 1; ReleasingSharedGPUs.erase(g); }
        }
    }
    bool isPipelinedToReleasingGpu(int64_t mem, int g) const {  // :237-245
        int64_t usedBefore = get(UsedSharedGPUsMemory, g) + mem, releasingBefore = get(ReleasingSharedGPUsMemory, g) - mem;
        bool usedOriginally0 = get(UsedSharedGPUsMemory, g) == 0, releasingOriginally0 = get(ReleasingSharedGPUsMemory, g) == 0;
        return usedBefore == releasingBefore || (usedOriginally0 && releasingOriginally0);
    }
    void removeSharedTaskResourcesPerPodGroup(int status, int64_t mem, int g) {  // :153-235
        UsedSharedGPUsMemory[g] -= mem;
        switch (status) {
            case KAI_POD_RELEASING:
                ReleasingSharedGPUsMemory[g] -= mem; AllocatedSharedGPUsMemory[g] -= mem;
                if (UsedSharedGPUsMemory[g] <= 0) {
                    if (int(GetNumberOfGPUsInNode()) >= int(Idle.gpus) + getNumberOfUsedGPUs()) Idle.gpus += 1;
                    if (isSharedGpuMarkedAsReleasing(g)) { Releasing.gpus -= 1; ReleasingSharedGPUs.erase(g); }
                }
                break;
            case KAI_POD_PIPELINED:
                ReleasingSharedGPUsMemory[g] += mem;
                if (isPipelinedToReleasingGpu(mem, g)) Releasing.gpus += 1;
                break;
            default:
                AllocatedSharedGPUsMemory[g] -= mem;
                if (UsedSharedGPUsMemory[g] <= 0) { if (int(GetNumberOfGPUsInNode()) >= int(Idle.gpus) + getNumberOfUsedGPUs()) Idle.gpus += 1; }
                if (isGpuReleasingFromSharedTasks(g) && !isSharedGpuMarkedAsReleasing(g)) { Releasing.gpus += 1; ReleasingSharedGPUs.insert(g); }
        }
    }
    bool AddTask(PodInfo* task) {  // :384-417
        setAcceptedResources(task);
        if (podInfos.count(task->idx)) return false;  // "task already on node"
        if (task->isLegacyMig) hasLegacyMigTasks = true;  // :407-409
        Resource r = task->accepted.AsResource();     // getAcceptedTaskResourceWithoutSharedGPU (gpu_sharing_node_info.go:52-66): no GPUs for a shared allocation
        const bool shared = task->IsSharedGPUAllocation();
        if (shared) r.gpus = 0;
        OnNode c{task->status, r, shared, task->gpuGroups, shared ? GetResourceGpuMemory(task->resReq) : 0};
        podInfos[task->idx] = c;
        addTaskResources(r, task->status);
        if (shared) for (int g : c.groups) addSharedTaskResourcesPerPodGroup(c.status, c.gpuMemory, g);  // addSharedTaskResources :68-81
        return true;
    }
    // ConsolidateSharedPodInfoToDifferentGPU (gpu_sharing_node_info.go:247-249 → addTask(ti, true), node_info.go:388-417): the node's copy of the
    // task (releasing on its old GPU group) is dropped from PodInfos WITHOUT taking its amounts back — they stay "being released" — and the
    // task is added again on its new group
    bool ConsolidateSharedPodInfoToDifferentGPU(PodInfo* task) {
        setAcceptedResources(task);
        if (task->IsSharedGPUAllocation()) podInfos.erase(task->idx); else if (podInfos.count(task->idx)) return false;
        Resource r = task->accepted.AsResource();
        const bool shared = task->IsSharedGPUAllocation();
        if (shared) r.gpus = 0;
        OnNode c{task->status, r, shared, task->gpuGroups, shared ? GetResourceGpuMemory(task->resReq) : 0};
        podInfos[task->idx] = c;
        addTaskResources(r, task->status);
        if (shared) for (int g : c.groups) addSharedTaskResourcesPerPodGroup(c.status, c.gpuMemory, g);
        return true;
    }
    bool RemoveTask(PodInfo* ti) {  // :495-513 — uses the node's copy, i.e. the status and groups at add time
        auto it = podInfos.find(ti->idx); if (it == podInfos.end()) return false;
        OnNode c = it->second; podInfos.erase(it);
        removeTaskResources(c.tracked, c.status);
        if (c.shared) for (int g : c.groups) removeSharedTaskResourcesPerPodGroup(c.status, c.gpuMemory, g);  // removeSharedTaskResources :138-151
        return true;
    }
    bool UpdateTask(PodInfo* ti) { if (!RemoveTask(ti)) return false; return AddTask(ti); }  // :571-576
};

// ---------------------------------------------------------------- plugins/proportion/resource_share/resource_share.go:12-21
struct ResourceShare {
    double Deserved = 0, FairShare = 0, MaxAllowed = 0, OverQuotaWeight = 0, Allocated = 0, AllocatedNotPreemptible = 0, Request = 0, Usage = 0;
    double GetRequestableShare() const { return MaxAllowed == KAI_UNLIMITED ? Request : std::fmin(MaxAllowed, Request); }  // :41-46
    double GetAllocatableShare() const {  // :51-61
        if (Deserved == KAI_UNLIMITED) return MaxAllowed;
        double allocatable = std::fmax(Deserved, FairShare);
        if (MaxAllowed != KAI_UNLIMITED) allocatable = std::fmin(MaxAllowed, allocatable);
        return allocatable;
    }
};
using ResourceQuantities = std::array<double, 3>;  // resource_quantities.go:18 (CPU, Memory, GPU)

// plugins/proportion/resource_share/queue_resource_share.go:21-29
struct QueueAttributes {
    int idx = -1; uint32_t uidRank = 0; int parent = -1; std::vector<int> children; int64_t createdNs = 0; int priority = 0;
    ResourceShare share[3];
    ResourceQuantities get(double ResourceShare::*f) const { return {share[0].*f, share[1].*f, share[2].*f}; }
    ResourceQuantities GetFairShare() const { return get(&ResourceShare::FairShare); }
    ResourceQuantities GetAllocatedShare() const { return get(&ResourceShare::Allocated); }
    ResourceQuantities GetDeservedShare() const { return get(&ResourceShare::Deserved); }
    ResourceQuantities GetAllocatableShare() const { return {share[0].GetAllocatableShare(), share[1].GetAllocatableShare(), share[2].GetAllocatableShare()}; }
    double GetDominantResourceShare(const ResourceQuantities& total) const {  // queue_resource_share.go:142-166
        double dominant = 0.0;
        for (int r = 0; r < 3; r++) {
            double value, allocatableShare = share[r].GetAllocatableShare();
            if (allocatableShare == KAI_UNLIMITED) allocatableShare = total[r];
            double allocated = share[r].Allocated;
            if (allocatableShare == 0) value = allocated * 1000; else value = allocated / allocatableShare;
            dominant = std::fmax(dominant, value);
        }
        return dominant;
    }
};

// resource_quantities.go:50-97
inline int compareQuantities(double q, double o) {
    if (q == KAI_UNLIMITED) return o == KAI_UNLIMITED ? 0 : 1;
    if (o == KAI_UNLIMITED) return -1;
    return q > o ? 1 : q < o ? -1 : 0;
}
inline bool rqLess(const ResourceQuantities& a, const ResourceQuantities& b) { for (int r = 0; r < 3; r++) if (a[r] >= b[r]) return false; return true; }
inline bool rqLessEqual(const ResourceQuantities& a, const ResourceQuantities& b) { for (int r = 0; r < 3; r++) if (compareQuantities(a[r], b[r]) > 0) return false; return true; }
inline bool rqLessInAtLeastOneResource(const ResourceQuantities& a, const ResourceQuantities& b) { return !rqLessEqual(b, a); }

// api/queue_info/queue_info.go:32-43
struct QueueInfo { int idx = -1; uint32_t uidRank = 0; int parent = -1; std::vector<int> children; int priority = 0; int64_t createdNs = 0;
                   int64_t preemptMinRuntimeNs = -1, reclaimMinRuntimeNs = -1;  // -1 = nil (queue_info.go PreemptMinRuntime / ReclaimMinRuntime)
                   bool IsLeafQueue() const { return children.empty(); } };

// ---------------------------------------------------------------- scheduler_util/priority_queue.go (container/heap semantics)
template <class T>
struct PriorityQueue {
    std::vector<T> items; std::function<bool(const T&, const T&)> lessFn; int maxQueueSize = -1;
    bool less(int i, int j) { return lessFn(items[i], items[j]); }
    void up(int j) { for (;;) { int i = (j - 1) / 2; if (i == j || !less(j, i)) break; std::swap(items[i], items[j]); j = i; } }
    bool down(int i0, int n) {
        int i = i0;
        for (;;) { int j1 = 2 * i + 1; if (j1 >= n || j1 < 0) break; int j = j1; int j2 = j1 + 1; if (j2 < n && less(j2, j1)) j = j2; if (!less(j, i)) break; std::swap(items[i], items[j]); i = j; }
        return i > i0;
    }
    T heapRemove(int i) { int n = int(items.size()) - 1; if (n != i) { std::swap(items[i], items[n]); if (!down(i, n)) up(i); } T x = items.back(); items.pop_back(); return x; }
    void Push(const T& x) {  // priority_queue.go:48-55
        items.push_back(x); up(int(items.size()) - 1);
        if (maxQueueSize != -1 && int(items.size()) > maxQueueSize) heapRemove(maxQueueSize);
    }
    T Pop() { int n = int(items.size()) - 1; std::swap(items[0], items[n]); down(0, n); T x = items.back(); items.pop_back(); return x; }
    const T& Peek() const { return items[0]; }
    void Fix(int i) { if (!down(i, int(items.size()))) up(i); }
    bool Empty() const { return items.empty(); }
    int Len() const { return int(items.size()); }
};

}  // namespace orc
