// oracle_solver.hpp — TEST INFRASTRUCTURE (see oracle_model.hpp header).
// Restates the victim search of the reference: actions/{reclaim,preempt,consolidation} on top of
// actions/common/solvers (JobSolver → PodAccumulatedScenarioBuilder → byPodSolver), the AccumulatedIdleGpus scenario
// filter, proportion's reclaimable validator and MinimalJobRepresentatives.  Included at the end of kai_oracle.cpp.
//
// Where the reference ranges over a Go map (random order) this restatement fixes ONE order — any order is a valid
// execution of the reference; the device engine follows the same choices (DESIGN.md "canonical orders"):
//   * jobs: ascending snapshot index;  nodes: ascending node name (name rank), "" (no node) first;  queues: ascending index
//   * pods of a job: pod-sets by name rank, pods by snapshot index inside a pod-set (PodGroupInfo::AllPods)
// The AccumulatedNodeAffinities scenario filter (accumulated_scenario_filters/node_affinities) is restated below and pinned on its own test table; it rejects a
// scenario only when some pending pod's node affinity matches none of the nodes the simulation may use — a necessary condition, so it prunes without changing
// results.  A session runs it only on request (kai_oracle_node_affinities_filter(1)): the snapshot carries the static Filters of a pod as ONE class table, not the
// node affinity by itself, see the struct's header.
#pragma once
#include <set>
#ifdef ORC_TRACE
#define ORC_T(...) fprintf(stderr, __VA_ARGS__)
#else
#define ORC_T(...) ((void)0)
#endif

namespace orc {

// ---------------------------------------------------------------- api/podgroup_info/job_info.go:477-510 CloneWithTasks
struct JobClone { PodGroupInfo job; std::vector<std::unique_ptr<PodSet>> sets; };
inline PodGroupInfo* Session::CloneWithTasks(PodGroupInfo* src, const std::vector<PodInfo*>& tasks) {
    auto c = std::make_unique<JobClone>();
    PodGroupInfo& j = c->job;
    j.idx = src->idx; j.uidRank = src->uidRank; j.queue = src->queue; j.priority = src->priority; j.preemptible = src->preemptible; j.createdNs = src->createdNs;
    j.rootGroup = src->rootGroup; j.isClone = true;
    for (auto* ps : src->podSets) {  // RootSubGroupSet.Clone(): pod-sets without pods, same minAvailable / constraints
        auto n = std::make_unique<PodSet>();
        n->idx = ps->idx; n->nameRank = ps->nameRank; n->job = ps->job; n->group = ps->group; n->tc = ps->tc; n->minAvailable = ps->minAvailable;
        j.podSets.push_back(n.get()); c->sets.push_back(std::move(n));
    }
    // the reference adds task.Clone(); statement operations mutate the object they are handed and re-index the ORIGINAL job with
    // it (statement.go:63-126 → job.UpdateTaskStatus), so "the task with this UID" is one object here
    for (auto* t : tasks) j.AddTaskInfo(t);
    PodGroupInfo* out = &c->job; clonePool.push_back(std::move(c));
    return out;
}

// ---------------------------------------------------------------- api/podgroup_info/eviction_info.go:14-97
inline std::vector<PodInfo*> Session::GetTasksToEvict(PodGroupInfo* job, bool& hasMoreTasks) {
    PriorityQueue<PodSet*> sgq; sgq.lessFn = [this](PodSet* const& l, PodSet* const& r) { return PodSetOrderFn(r, l); };
    for (auto* ps : job->podSets) sgq.Push(ps);
    int maxNumOfSubGroups = int(job->podSets.size());  // getNumOfSubGroupsToEvict :49-59
    for (auto* ps : job->podSets) if (ps->numActiveAllocatedTasks > int(ps->minAvailable)) { maxNumOfSubGroups = 1; break; }
    std::vector<PodInfo*> tasksToEvict; int numEvictedSubGroups = 0;
    while (!sgq.Empty() && numEvictedSubGroups < maxNumOfSubGroups) {
        PodSet* next = sgq.Pop();
        PriorityQueue<PodInfo*> tq; tq.lessFn = [this](PodInfo* const& l, PodInfo* const& r) { return TaskOrderFn(r, l); };
        for (auto& kv : next->podInfos) if (IsActiveAllocatedStatus(kv.second->status)) tq.Push(kv.second);
        int maxTasksToEvict = next->numActiveAllocatedTasks > int(next->minAvailable) ? 1 : next->numActiveAllocatedTasks;  // :61-67
        int n = 0; while (!tq.Empty() && n < maxTasksToEvict) { tasksToEvict.push_back(tq.Pop()); n++; }
        numEvictedSubGroups += 1;
    }
    int numAllocatedTasks = 0; for (auto* t : job->AllPods()) if (IsActiveAllocatedStatus(t->status)) numAllocatedTasks++;  // GetActiveAllocatedTasksCount
    hasMoreTasks = int(tasksToEvict.size()) < numAllocatedTasks;
    return tasksToEvict;
}

// ---------------------------------------------------------------- solvers/scenario/{base_scenario,by_node_scenario}.go
struct VictimInfo { PodGroupInfo* Job = nullptr; std::vector<PodInfo*> Tasks; };
struct Scenario {
    Session* ssn; PodGroupInfo* preemptor;
    std::map<int, VictimInfo> victims;                                 // by job index
    std::vector<PodInfo*> pendingTasks, potentialVictimsTasks, recordedVictimsTasks;
    std::vector<PodGroupInfo*> recordedVictimsJobs;
    std::map<int, std::vector<PodGroupInfo*>> victimsJobsTaskGroups;   // job index → representatives (clones)
    std::map<int, std::vector<int>> potentialVictimsJobsByNode;        // node index (-1 = no node) → job indices, first-seen order

    Scenario() : ssn(nullptr), preemptor(nullptr) {}  // (task lists filled by hand: kai_oracle_idle_gpus_kat)
    Scenario(Session* s, PodGroupInfo* pendingJob, const std::vector<PodGroupInfo*>& recorded) : ssn(s), preemptor(pendingJob) {  // base_scenario.go:35-71
        pendingTasks = pendingJob->AllPods();
        for (auto* rj : recorded) { recordedVictimsJobs.push_back(rj); appendTasksAsVictimJob(rj->AllPods()); }
        for (auto* rj : recordedVictimsJobs) for (auto* t : rj->AllPods()) recordedVictimsTasks.push_back(t);
    }
    PodGroupInfo* LatestPotentialVictim() const { return potentialVictimsTasks.empty() ? nullptr : &ssn->jobs[potentialVictimsTasks.back()->job]; }  // :97-103
    void appendTasksAsVictimJob(const std::vector<PodInfo*>& tasks) {  // :118-137
        PodGroupInfo* originalJob = &ssn->jobs[tasks[0]->job];
        PodGroupInfo* job = ssn->CloneWithTasks(originalJob, tasks);
        victimsJobsTaskGroups[job->idx].push_back(job);
        VictimInfo& v = victims[job->idx]; v.Job = originalJob;
        v.Tasks.insert(v.Tasks.end(), tasks.begin(), tasks.end());
    }
    void AddPotentialVictimsTasks(const std::vector<PodInfo*>& tasks) {  // base :109-116 + by_node :47-53
        if (tasks.empty()) return;
        potentialVictimsTasks.insert(potentialVictimsTasks.end(), tasks.begin(), tasks.end());
        appendTasksAsVictimJob(tasks);
        for (auto* t : tasks) { auto& l = potentialVictimsJobsByNode[t->node]; if (std::find(l.begin(), l.end(), t->job) == l.end()) l.push_back(t->job); }
    }
    std::vector<PodInfo*> VictimsTasksFromNodes(const std::vector<int>& nodes) {  // by_node_scenario.go:58-80 (the reference collects the jobs in a Go map: any order; here first seen)
        std::vector<PodInfo*> tasks; std::vector<int> victimsJobs;
        for (int node : nodes) {
            auto it = potentialVictimsJobsByNode.find(node); if (it == potentialVictimsJobsByNode.end()) continue;
            for (int jobID : it->second) if (std::find(victimsJobs.begin(), victimsJobs.end(), jobID) == victimsJobs.end()) victimsJobs.push_back(jobID);
        }
        for (int jobID : victimsJobs) for (auto* group : victimsJobsTaskGroups[jobID]) for (auto* t : group->AllPods()) tasks.push_back(t);
        return tasks;
    }
    std::vector<PodInfo*> VictimsTasksFromNodes(int node) { return VictimsTasksFromNodes(std::vector<int>{node}); }  // by_pod_solver.go:169 asks about one node
    PodGroupInfo* GetVictimJobRepresentativeById(PodInfo* victim) {  // base_scenario.go:139-149
        for (auto* rep : victimsJobsTaskGroups[victim->job]) for (auto* t : rep->AllPods()) if (t->idx == victim->idx) return rep;
        return nullptr;
    }
};

// ---------------------------------------------------------------- accumulated_scenario_filters/idle_gpus/{idle_gpus,common}.go
// accumulated_scenario_filters/idle_gpus/common.go:34-64: every requirement (sorted descending) is matched to the first holder (sorted descending by capacity) that
// still has room after what earlier requirements took from it; a zero requirement ends the list
template <class Reqs, class Holders, class Cap>
inline bool greedyMatchRequirements(const Reqs& requirements, const Holders& holders, Cap capacity) {
    std::map<int, double> virtuallyAllocated;
    for (double required : requirements) {
        if (required == 0) return true;
        bool matched = false;
        for (int holder : holders) {
            const double totalCapacity = capacity(holder);
            if (totalCapacity < required) break;  // holders are sorted: nobody behind this one can take it either
            if (totalCapacity - virtuallyAllocated[holder] >= required) { virtuallyAllocated[holder] += required; matched = true; break; }
        }
        if (!matched) return false;
    }
    return true;
}

struct AccumulatedIdleGpus {
    std::vector<double> requiredGpusSorted;
    std::map<int, double> nodesNameToIdleGpus;
    std::vector<int> maxFreeGpuNodesSorted;
    std::set<int> pendingTasksInState, recordedVictimsInCache, potentialVictimsInCache;
    bool valid = true;

    static int cmpDesc(const std::map<int, double>& m, int elem, int target) {  // cmp.Compare(idle[target], idle[elem])
        double a = m.at(target), b = m.at(elem); return a < b ? -1 : a > b ? 1 : 0;
    }
    void orderedInsert(int t, bool replace) {  // idle_gpus.go:198-247 (slices.BinarySearchFunc + in-place shifts)
        auto& ts = maxFreeGpuNodesSorted;
        int lo = 0, hi = int(ts.size());
        while (lo < hi) { int h = (lo + hi) / 2; if (cmpDesc(nodesNameToIdleGpus, ts[h], t) < 0) lo = h + 1; else hi = h; }
        int i = lo;
        for (int j = 0; j < int(ts.size()); j++) if (ts[j] == t) {  // updateLocationIfTAlreadyExists
            if (j == i) return;
            if (i < j) { int elem = ts[j]; for (int k = j; k > i; k--) ts[k] = ts[k - 1]; ts[i] = elem; }  // shiftElementLeft
            return;
        }
        if (replace) {  // insertWithoutIncreasingListSize
            if (i == int(ts.size()) - 1) { ts[i] = t; return; }
            for (int j = int(ts.size()) - 1; j > i; j--) ts[j] = ts[j - 1];
            if (i < int(ts.size())) ts[i] = t;
        } else ts.insert(ts.begin() + i, t);
    }
    AccumulatedIdleGpus() = default;  // (the filter's fields set by hand: kai_oracle_idle_gpus_kat, the cases of idle_gpus_test.go)
    AccumulatedIdleGpus(Session* ssn, Scenario* sc) {  // NewIdleGpusFilter :53-71 + createGpuMap :176-196
        int relevantNodesLen = int(sc->pendingTasks.size()); double minRelevantValue = -1;
        for (auto& ni : ssn->nodes) {
            nodesNameToIdleGpus[ni.idx] = ni.GetSumOfIdleGPUs() + ni.GetSumOfReleasingGPUs();  // nodeIdleOrReleasingGpuCapacity (idle_gpus/common.go:12-16)
            if (nodesNameToIdleGpus[ni.idx] > minRelevantValue || int(maxFreeGpuNodesSorted.size()) < relevantNodesLen) {
                bool replace = relevantNodesLen <= int(maxFreeGpuNodesSorted.size());
                orderedInsert(ni.idx, replace);
                if (!maxFreeGpuNodesSorted.empty()) minRelevantValue = nodesNameToIdleGpus[maxFreeGpuNodesSorted.back()];
            }
        }
        valid = updateStateWithScenario(sc, true);
    }
    bool Filter(Scenario* sc, bool& err) {  // :77-88
        err = !updateStateWithScenario(sc, false);
        if (err) return false;
        return greedyMatchRequirements(requiredGpusSorted, maxFreeGpuNodesSorted, [&](int holder) { return nodesNameToIdleGpus[holder]; });
    }
    int updateVictimList(const std::vector<PodInfo*>& victimTasks, std::set<int>& cache) {  // :111-122 + iterateNewVictims common.go:66-88
        int minIdleGpusRelevant = maxFreeGpuNodesSorted.empty() ? -2 : maxFreeGpuNodesSorted.back(); int hits = 0;
        for (auto* task : victimTasks) {
            if (task->node < 0) continue;
            if (cache.count(task->idx)) { hits++; continue; }
            cache.insert(task->idx);
            // updateWithVictim :160-174
            if (nodesNameToIdleGpus.empty()) { minIdleGpusRelevant = -2; continue; }
            double prevMinRelevantValue = minIdleGpusRelevant == -2 ? 0.0 : nodesNameToIdleGpus[minIdleGpusRelevant];
            nodesNameToIdleGpus[task->node] += task->accepted.GPUs();
            if (nodesNameToIdleGpus[task->node] > prevMinRelevantValue) {
                orderedInsert(task->node, true);
                minIdleGpusRelevant = maxFreeGpuNodesSorted.empty() ? -2 : maxFreeGpuNodesSorted.back();
            }
        }
        return hits;
    }
    bool updateStateWithScenario(Scenario* sc, bool isFirstScenario) {  // :90-109, 124-158
        if (!isFirstScenario) for (auto* pod : sc->pendingTasks) if (!pendingTasksInState.count(pod->idx)) return false;
        std::vector<double> required;
        for (auto* pod : sc->pendingTasks) { required.push_back(pod->resReq.GPUs()); pendingTasksInState.insert(pod->idx); }
        std::sort(required.begin(), required.end(), [](double a, double b) { return a > b; });
        requiredGpusSorted = required;
        size_t pre = recordedVictimsInCache.size();
        size_t hits = size_t(updateVictimList(sc->recordedVictimsTasks, recordedVictimsInCache));
        if (!(isFirstScenario || (pre == hits && hits == recordedVictimsInCache.size()))) return false;
        size_t preP = potentialVictimsInCache.size();
        size_t hitsP = size_t(updateVictimList(sc->potentialVictimsTasks, potentialVictimsInCache));
        if (preP != hitsP) return false;
        return true;
    }
};

// ---------------------------------------------------------------- accumulated_scenario_filters/idle_gpus/topology_aware_idle_gpus.go
// Per (topology, required level) of the preemptor's SubGroupSets: the GPU totals of those sub-groups, largest first, are matched
// greedily against the level's domains by idle + releasing + freed GPUs.  With two sub-groups on one level the greedy match can
// reject a feasible scenario, so the filter is part of the result, not only a shortcut.  (The reference keys a domain by the node's
// label VALUE at that level; here a domain is the snapshot's domain of that level row — the same thing when label values are unique
// across parents.)
struct TopologyAwareIdleGpus {
    Session* ssn; bool active = false;
    std::vector<std::pair<int, int>> subgroups;          // (SubGroupSet idx, level row) of the preemptor's groups with a required level
    std::vector<int> rows;                                // distinct level rows (constraint keys)
    std::map<int, double> domainCapacity;                 // domain → idle GPUs
    std::map<int, std::vector<int>> domainsByRow;         // level row → domains, capacity descending
    std::set<int> processedVictims;
    TopologyAwareIdleGpus(Session* s, Scenario* sc) : ssn(s) {
        if (ssn->nTopologies == 0) return;
        std::function<void(int)> walk = [&](int g) {  // getSubgroupsWithRequiredConstraints :175-189 (SubGroupSets only)
            const SubGroupSet& sg = ssn->groups[g];
            if (sg.tc.topology >= 0 && sg.tc.required >= 0 && sg.tc.required < ssn->topoLevelOff[sg.tc.topology + 1] - ssn->topoLevelOff[sg.tc.topology])
                subgroups.push_back({g, ssn->topoLevelOff[sg.tc.topology] + sg.tc.required});
            for (int c : sg.groups) walk(c);
        };
        walk(sc->preemptor->rootGroup);
        if (subgroups.empty()) return;
        active = true;
        for (auto& sr : subgroups) if (std::find(rows.begin(), rows.end(), sr.second) == rows.end()) rows.push_back(sr.second);
        const int N = int(ssn->nodes.size());
        for (auto& ni : ssn->nodes) {  // buildDomainCapacity :191-244
            double total = ni.GetSumOfIdleGPUs() + ni.GetSumOfReleasingGPUs();
            for (int row : rows) { int d = ssn->nodeDomain[size_t(row) * N + ni.idx]; if (d < 0) continue; domainCapacity[d] += total; }
        }
        for (int row : rows) {
            std::vector<int>& v = domainsByRow[row];
            for (auto& kv : domainCapacity) if (ssn->domains[kv.first].level + ssn->topoLevelOff[ssn->domains[kv.first].topo] == row) v.push_back(kv.first);
            std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return domainCapacity[a] > domainCapacity[b]; });
        }
    }
    void applyVictimTasks(const std::vector<PodInfo*>& tasks) {  // :73-97
        const int N = int(ssn->nodes.size());
        for (auto* t : tasks) {
            if (t->node < 0 || processedVictims.count(t->idx)) continue;
            processedVictims.insert(t->idx);
            double freed = t->accepted.GPUs();
            for (int row : rows) {
                int d = ssn->nodeDomain[size_t(row) * N + t->node]; if (d < 0) continue;
                domainCapacity[d] += freed;
                std::vector<int>& v = domainsByRow[row];  // repositionDomainAfterIncrease :99-128
                int cur = -1; for (int i = 0; i < int(v.size()); i++) if (v[i] == d) { cur = i; break; }
                if (cur <= 0) continue;
                int np = cur; while (np > 0 && domainCapacity[v[np - 1]] < domainCapacity[d]) np--;
                if (np != cur) { int e = v[cur]; for (int k = cur; k > np; k--) v[k] = v[k - 1]; v[np] = e; }
            }
        }
    }
    bool Filter(Scenario* sc) {  // :66-71 + requiredTopologyCapacityExists :130-147
        applyVictimTasks(sc->recordedVictimsTasks); applyVictimTasks(sc->potentialVictimsTasks);
        for (int row : rows) {
            std::vector<double> req;
            for (auto& sr : subgroups) if (sr.second == row) {  // sumGpuRequirements :246-257 over the representative's pods
                std::vector<PodSet*> under; ssn->allPodSets(sc->preemptor, &ssn->groups[sr.first], under);
                double sum = 0; for (auto* ps : under) for (auto& kv : ps->podInfos) sum += kv.second->resReq.GPUs();
                req.push_back(sum);
            }
            std::sort(req.begin(), req.end(), [](double a, double b) { return a > b; });
            if (!greedyMatchRequirements(req, domainsByRow[row], [&](int d) { return domainCapacity[d]; })) return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------- accumulated_scenario_filters/node_affinities/node_affinities.go
// The filter keeps the set of nodes a simulation may use — the feasible nodes it was created with plus the node of every victim it has seen (:100-121) — and
// rejects a scenario when a pending pod with a REQUIRED node affinity (a node selector or requiredDuringScheduling terms, :134-145) has no node there that
// its affinity matches (:123-132).  "Matches" is the upstream NodeAffinity plugin (k8s.io/kubernetes v1.34.2, pkg/scheduler/framework/plugins/nodeaffinity —
// a go.mod dependency, not under /root/reference), asked in two steps (:147-166): PreFilter may narrow the candidates to the node NAMES the pod's terms list
// (every term carries a metadata.name matchFields requirement; these are looked up among ALL nodes of the cluster, feasible or not, :157-160), no narrowing =
// the filter's own node set (:168-188); Filter then decides per node.  The three questions come in as callbacks: the known-answer entry
// (kai_oracle_node_affinities_kat) answers them from the test table's labels and pod specs; a session answers them from the static class table
// (class_fit: every static Filter of the pod, the node affinity among them — a pod can only land on a node of its class row, so the condition stays a necessary
// one; it prunes at least what the reference's filter prunes and changes no result: tests/test_oracle_kat.py runs the victim actions with and without it).
struct AccumulatedNodeAffinities {
    std::set<int> feasibleNodes, processedVictims;   // node indices; pod indices (:32-36; allNodes / allNodeInfos = the callbacks' business)
    std::function<bool(const PodInfo*)> hasRequiredNodeAffinity;                              // :134-145
    std::function<int(const PodInfo*, std::vector<int>&)> preFilter;                          // 0 = no narrowing (nil result or Skip), 1 = names (cluster node indices; absent nodes left out), -1 = unschedulable
    std::function<bool(const PodInfo*, int)> filter;                                          // NodeAffinity.Filter(pod, node)

    // NewNodeAffinitiesFilter :38-59: nullptr without a scenario or without a pending pod that has a required affinity
    static std::unique_ptr<AccumulatedNodeAffinities> create(Scenario* sc, const std::set<int>& feasible, std::function<bool(const PodInfo*)> hasRequired,
                                                             std::function<int(const PodInfo*, std::vector<int>&)> pre, std::function<bool(const PodInfo*, int)> flt) {
        if (!sc) return nullptr;
        bool any = false; for (auto* t : sc->pendingTasks) if (hasRequired(t)) { any = true; break; }  // preemptorHasPodsWithNodeAffinities :72-79
        if (!any) return nullptr;
        auto f = std::make_unique<AccumulatedNodeAffinities>();
        f->feasibleNodes = feasible; f->hasRequiredNodeAffinity = std::move(hasRequired); f->preFilter = std::move(pre); f->filter = std::move(flt);
        f->updateStateWithScenario(sc);
        return f;
    }
    void updateVictimNodesFromTask(const PodInfo* task) {  // :100-112
        if (!processedVictims.insert(task->idx).second) return;
        if (task->node < 0) return;
        feasibleNodes.insert(task->node);
    }
    void updateStateWithScenario(Scenario* sc) {  // :91-98
        for (auto* t : sc->potentialVictimsTasks) updateVictimNodesFromTask(t);
        for (auto* t : sc->recordedVictimsTasks) updateVictimNodesFromTask(t);
    }
    bool hasNodeMatchingPodInSet(const PodInfo* task) {  // :147-166 + preFilteredNodeNames :168-188
        std::vector<int> names; const int st = preFilter(task, names);
        if (st < 0) return false;
        if (st == 0) names.assign(feasibleNodes.begin(), feasibleNodes.end());
        for (int n : names) if (filter(task, n)) return true;
        return false;
    }
    bool Filter(Scenario* sc) {  // :85-89
        updateStateWithScenario(sc);
        for (auto* t : sc->pendingTasks) { if (!hasRequiredNodeAffinity(t)) continue; if (!hasNodeMatchingPodInSet(t)) return false; }  // allPendingPodsHaveMatchingNodes :114-123
        return true;
    }
};
inline int g_node_affinities_filter = 0;   // kai_oracle_node_affinities_filter: sessions run the filter on the class table (0 = as before: not at all)
inline int64_t g_node_affinities_dropped = 0;  // scenarios it dropped since the switch was last set

// ---------------------------------------------------------------- solvers/pod_scenario_builder.go
struct ScenarioBuilder {
    Session* ssn; std::unique_ptr<Scenario> lastScenario; std::unique_ptr<AccumulatedIdleGpus> idleGpus; std::unique_ptr<TopologyAwareIdleGpus> topoGpus;
    std::unique_ptr<AccumulatedNodeAffinities> nodeAffinities;
    JobsOrderByQueues* victimsJobsQueue; std::set<int> recordedVictimsTasks;
    ScenarioBuilder(Session* s, PodGroupInfo* pendingJob, const std::vector<PodGroupInfo*>& recordedVictimsJobs, JobsOrderByQueues* vq, const std::set<int>& feasibleNodes) : ssn(s), victimsJobsQueue(vq) {  // :32-76
        if (!ssn->GetTasksToAllocate(pendingJob, false).empty()) {
            lastScenario = std::make_unique<Scenario>(ssn, pendingJob, recordedVictimsJobs);
            for (auto* job : recordedVictimsJobs) for (auto* t : job->AllPods()) recordedVictimsTasks.insert(t->idx);
            if (g_node_affinities_filter && !ssn->classFit.empty()) {  // :52-56, on the class table (see AccumulatedNodeAffinities)
                Session* S = ssn;
                auto fits = [S](const PodInfo* t, int n) { return S->classFit[size_t(t->podClass) * S->nNodeClasses + S->nodes[n].nodeClass] != 0; };
                auto required = [S, fits](const PodInfo* t) { for (int n = 0; n < (int)S->nodes.size(); n++) if (!fits(t, n)) return true; return false; };
                nodeAffinities = AccumulatedNodeAffinities::create(lastScenario.get(), feasibleNodes, required, [](const PodInfo*, std::vector<int>&) { return 0; }, fits);
            }
            topoGpus = std::make_unique<TopologyAwareIdleGpus>(ssn, lastScenario.get());
            if (!topoGpus->active) topoGpus.reset();
            idleGpus = std::make_unique<AccumulatedIdleGpus>(ssn, lastScenario.get());
            if (!idleGpus->valid) idleGpus.reset();
        }
    }
    bool addNextPotentialVictims() {  // :91-133
        PodGroupInfo* nextVictimJob = victimsJobsQueue->PopNextJob();
        ORC_T("[orc] victim pop %d (preemptor %d)\n", nextVictimJob->idx, lastScenario ? lastScenario->preemptor->idx : -1);
        bool jobHasMoreTasks = false;
        std::vector<PodInfo*> potentialVictimTasks = ssn->GetTasksToEvict(nextVictimJob, jobHasMoreTasks);
        for (auto* pv : potentialVictimTasks) if (recordedVictimsTasks.count(pv->idx)) {
            std::vector<PodInfo*> remaining; for (auto* t : nextVictimJob->AllPods()) if (!recordedVictimsTasks.count(t->idx)) remaining.push_back(t);
            if (!remaining.empty()) victimsJobsQueue->PushJob(ssn->CloneWithTasks(nextVictimJob, remaining));
            return false;
        }
        if (jobHasMoreTasks) {
            std::vector<PodInfo*> remaining;
            for (auto* t : nextVictimJob->AllPods()) if (std::find(potentialVictimTasks.begin(), potentialVictimTasks.end(), t) == potentialVictimTasks.end()) remaining.push_back(t);
            victimsJobsQueue->PushJob(ssn->CloneWithTasks(nextVictimJob, remaining));
        }
        if (lastScenario) lastScenario->AddPotentialVictimsTasks(potentialVictimTasks);
        return true;
    }
    Scenario* GetNextScenario() {  // :78-89
        for (;;) {
            if (victimsJobsQueue->IsEmpty()) return nullptr;
            if (!addNextPotentialVictims()) continue;
            return GetValidScenario();
        }
    }
    Scenario* GetValidScenario() {  // :135-161
        bool valid = true;
        if (nodeAffinities && lastScenario && !nodeAffinities->Filter(lastScenario.get())) { valid = false; g_node_affinities_dropped++; }
        if (valid && topoGpus && lastScenario && !topoGpus->Filter(lastScenario.get())) valid = false;
        if (valid && idleGpus && lastScenario) { bool err = false; bool ok = idleGpus->Filter(lastScenario.get(), err); if (!err && !ok) valid = false; }
        if (!valid) { ssn->stats.scenariosFiltered++; return GetNextScenario(); }
        return lastScenario.get();
    }
};

// ---------------------------------------------------------------- solvers/by_pod_solver.go + common/action.go
struct SolutionResult { bool solved = false; std::vector<PodInfo*> victimsTasks; std::vector<PodGroupInfo*> victimJobs; std::unique_ptr<Statement> statement; };
using SolutionValidator = std::function<bool(Scenario*)>;

struct ByPodSolver {
    Session* ssn; std::set<int>& feasibleNodes; SolutionValidator validator; bool allowVictimConsolidation;
    enum Sim { simNone, simSolved, simRejected };

    void EvictAllPreemptees(Statement& stmt, const std::vector<PodInfo*>& tasks) { for (auto* t : tasks) stmt.Evict(t); }  // common/action.go:27-52
    // common/action.go:65-122 (GetJobsToAllocate + TryToVirtuallyAllocatePreemptorAndGetVictims)
    bool tryScenarioWithEvictedVictims(Scenario* sc, Statement& stmt, const std::vector<PodInfo*>& victimTasks) {  // by_pod_solver.go:211-237
        PodGroupInfo* pendingJob = sc->preemptor;
        std::vector<NodeInfo*> nodes; for (int n : feasibleNodes) nodes.push_back(&ssn->nodes[n]);
        std::map<int, PodGroupInfo*> all;  // keyed by job index: the preemptor's representative replaces the session's job of the same UID
        for (auto& job : ssn->jobs) if (job.GetNumPendingTasks() > 0) all[job.idx] = &job;  // utils.GetAllPendingJobs
        std::set<int> potentialVictims;
        for (auto* t : victimTasks) { all[t->job] = &ssn->jobs[t->job]; potentialVictims.insert(t->job); }
        all[pendingJob->idx] = pendingJob;
        JobsOrderInitOptions o; o.MaxJobsQueueDepth = -1;
        JobsOrderByQueues jobsToAllocate(ssn, o);
        std::vector<PodGroupInfo*> v; for (auto& kv : all) v.push_back(kv.second);
        jobsToAllocate.InitializeWithJobs(v);
        bool preemptorAllocated = false;
        while (!jobsToAllocate.IsEmpty()) {
            PodGroupInfo* job = jobsToAllocate.PopNextJob();
            ORC_T("[orc] ja pop %d\n", job->idx);
            if (!potentialVictims.count(job->idx) && job->idx != pendingJob->idx) continue;
            ssn->GetTasksToAllocateInitResource(job, false);  // evaluated for a log line; fills the job's cache like the reference does
            if (job->idx != pendingJob->idx) { ssn->AllocateJob(stmt, nodes, job, true); continue; }
            if (!ssn->AllocateJob(stmt, nodes, job, true)) return false;
            preemptorAllocated = true;
        }
        return preemptorAllocated;
    }
    Sim runSimulation(Scenario* sc, std::unique_ptr<Statement>& stmt, const std::vector<PodInfo*>& victimTasks, SolutionResult& out) {  // :100-116
        ssn->stats.simulations++;
        { bool okk = tryScenarioWithEvictedVictims(sc, *stmt, victimTasks); ORC_T("[orc] sim preemptor %d nvt %d -> %d   ops_len %d\n", sc->preemptor->idx, (int)victimTasks.size(), (int)okk, (int)stmt->operations.size()); if (!okk) return simNone; }
        std::vector<PodInfo*> preempted, pipelined;
        for (auto* t : victimTasks) { if (t->status == Releasing) preempted.push_back(t); else if (t->status == Pipelined) pipelined.push_back(t); }
        // handleScenarioSolution :171-197
        std::vector<PodInfo*> victimsTasks = preempted;
        if (!allowVictimConsolidation) victimsTasks.insert(victimsTasks.end(), pipelined.begin(), pipelined.end());
        std::vector<PodGroupInfo*> victimJobs = getVictimJobsFromVictimTasks(victimsTasks, sc);
        { bool v = !validator || validator(sc); ORC_T("[orc] validator -> %d\n", (int)v); if (!v) { stmt->Discard(); out = SolutionResult{}; return simRejected; } }
        if (allowVictimConsolidation) { victimsTasks.insert(victimsTasks.end(), pipelined.begin(), pipelined.end()); victimJobs = getVictimJobsFromVictimTasks(victimsTasks, sc); }
        out.solved = true; out.victimsTasks = victimsTasks; out.victimJobs = victimJobs; out.statement = std::move(stmt);
        return simSolved;
    }
    static std::vector<PodGroupInfo*> getVictimJobsFromVictimTasks(const std::vector<PodInfo*>& tasks, Scenario* sc) {  // :239-276
        std::map<int, std::vector<PodGroupInfo*>> jobs;
        for (auto* task : tasks) {
            bool exists = false;
            auto it = jobs.find(task->job);
            if (it != jobs.end()) for (auto* dup : it->second) { for (auto* p : dup->AllPods()) if (p->idx == task->idx) { exists = true; break; } if (exists) break; }
            if (!exists) { PodGroupInfo* m = sc->GetVictimJobRepresentativeById(task); if (m) jobs[m->idx].push_back(m); }
        }
        std::vector<PodGroupInfo*> out; for (auto& kv : jobs) out.insert(out.end(), kv.second.begin(), kv.second.end());
        return out;
    }
    SolutionResult solve(Scenario* sc) {  // :61-98
        auto stmt = std::make_unique<Statement>(ssn);
        SolutionResult res;
        EvictAllPreemptees(*stmt, sc->recordedVictimsTasks);
        PodGroupInfo* latest = sc->LatestPotentialVictim();
        if (!latest) {
            if (!sc->recordedVictimsTasks.empty()) { Sim s = runSimulation(sc, stmt, sc->recordedVictimsTasks, res); if (s != simNone) return res; }
        } else {
            // getNodesOfJob :199-209: the distinct NodeName of every pod of the job ("" included) — maps.Keys, random in the reference; canonical: by node name, "" first
            std::vector<int> nodeNames; for (auto* t : latest->AllPods()) if (std::find(nodeNames.begin(), nodeNames.end(), t->node) == nodeNames.end()) nodeNames.push_back(t->node);
            std::sort(nodeNames.begin(), nodeNames.end(), [&](int a, int b) { if (a < 0 || b < 0) return a < b; return ssn->nodes[a].nameRank < ssn->nodes[b].nameRank; });
            for (int nodeToTest : nodeNames) {  // solveOnPotentialNodes :118-144
                int cp = stmt->Checkpoint();
                std::vector<PodInfo*> potential = sc->VictimsTasksFromNodes(nodeToTest);
                EvictAllPreemptees(*stmt, potential);
                std::vector<int> newFeasible;
                for (auto* t : potential) if (t->node >= 0 && feasibleNodes.insert(t->node).second) newFeasible.push_back(t->node);
                std::vector<PodInfo*> victimTasks = sc->recordedVictimsTasks; victimTasks.insert(victimTasks.end(), potential.begin(), potential.end());
                Sim s = runSimulation(sc, stmt, victimTasks, res);
                if (s != simNone) return res;
                for (int n : newFeasible) feasibleNodes.erase(n);
                stmt->Rollback(cp);
            }
        }
        stmt->Discard();
        return SolutionResult{};
    }
};

// ---------------------------------------------------------------- solvers/job_solver.go
struct JobSolver {
    Session* ssn; std::vector<int> feasibleNodes; SolutionValidator validator; std::function<std::unique_ptr<JobsOrderByQueues>()> generateVictimsQueue;

    static PodGroupInfo* getPartialJobRepresentative(Session* ssn, PodGroupInfo* job, const std::vector<PodInfo*>& pendingTasks) {  // :128-151
        PodGroupInfo* rep = ssn->CloneWithTasks(job, pendingTasks);
        std::map<int, int> minAvailable; for (auto* t : pendingTasks) minAvailable[t->podset] += 1;
        for (auto& kv : minAvailable) for (auto* ps : rep->podSets) if (ps->idx == kv.first) ps->minAvailable = int32_t(kv.second);
        return rep;
    }
    bool Solve(PodGroupInfo* pendingJob, std::unique_ptr<Statement>& statementOut) {  // :50-93
        std::vector<PodGroupInfo*> recordedVictimsJobs; std::vector<PodInfo*> recordedVictimsTasks;
        int originalNumActiveTasks = 0; for (auto* ps : pendingJob->podSets) originalNumActiveTasks += ps->numActiveUsedTasks;
        ORC_T("[orc] solve job %d cached %d pending %d\n", pendingJob->idx, (int)pendingJob->hasTasksToAllocate, pendingJob->GetNumPendingTasks());
        std::vector<PodInfo*> tasksToAllocate = ssn->GetTasksToAllocate(pendingJob, false), pendingTasks;
#ifdef ORC_TRACE
        ORC_T("[orc] solve job %d ntasks %d\n", pendingJob->idx, (int)tasksToAllocate.size()); for (auto* t : tasksToAllocate) ORC_T("[orc]    task %d status %d virtual %d\n", t->idx, t->status, (int)t->isVirtualStatus); for (auto* t : pendingJob->AllPodsByIndex()) ORC_T("[orc]    pod %d status %d virtual %d node %d\n", t->idx, t->status, (int)t->isVirtualStatus, t->node);
#endif
        for (auto* next : tasksToAllocate) {
            pendingTasks.push_back(next);
            bool satisfactory = pendingTasks.size() == tasksToAllocate.size();
            PodGroupInfo* partial = getPartialJobRepresentative(ssn, pendingJob, pendingTasks);
            SolutionResult result = solvePartialJob(recordedVictimsJobs, recordedVictimsTasks, partial);
            ORC_T("[orc] partial result solved %d\n", (int)result.solved);
            if (!result.solved) break;
            if (!satisfactory && result.statement) result.statement->Discard();
            statementOut = std::move(result.statement);
            recordedVictimsTasks = result.victimsTasks; recordedVictimsJobs = result.victimJobs;
        }
        int numActiveTasks = 0; for (auto* ps : pendingJob->podSets) numActiveTasks += ps->numActiveUsedTasks;
        bool jobSolved = pendingJob->IsGangSatisfied();
        if (originalNumActiveTasks >= numActiveTasks) jobSolved = false;
        return jobSolved;
    }
    SolutionResult solvePartialJob(const std::vector<PodGroupInfo*>& recordedVictimsJobs, const std::vector<PodInfo*>& recordedVictimsTasks, PodGroupInfo* partial) {  // :95-126
        std::set<int> feasibleNodeMap(feasibleNodes.begin(), feasibleNodes.end());
        for (auto* t : recordedVictimsTasks) if (t->node >= 0) feasibleNodeMap.insert(t->node);
        std::unique_ptr<JobsOrderByQueues> vq = generateVictimsQueue();
        ScenarioBuilder builder(ssn, partial, recordedVictimsJobs, vq.get(), feasibleNodeMap);
        ORC_T("[orc] partial job %d: victims queue len %d empty %d scenario %d feasible %d\n", partial->idx, vq->Len(), (int)vq->IsEmpty(), (int)(builder.lastScenario != nullptr), (int)feasibleNodeMap.size());
        for (Scenario* sc = builder.GetValidScenario(); sc; sc = builder.GetNextScenario()) {
            ByPodSolver solver{ssn, feasibleNodeMap, validator, ssn->cfg.allow_consolidating_reclaim != 0};
            ssn->stats.scenarios++;
            SolutionResult r = solver.solve(sc);
            if (r.solved) return r;
        }
        return SolutionResult{};
    }
};

// ---------------------------------------------------------------- common/feasible_nodes.go
inline std::vector<int> Session::FeasibleNodesForJob(PodGroupInfo* job) {
    std::vector<int> out;
    bool allNeedGpu = true; for (auto* t : job->AllPods()) if (t->IsCPUOnlyRequest()) { allNeedGpu = false; break; }  // !IsRequireAnyKindOfGPU
    for (auto& n : nodes) if (!allNeedGpu || n.GetSumOfIdleGPUs() > 0 || n.GetSumOfReleasingGPUs() > 0) out.push_back(n.idx);
    return out;
}

// ---------------------------------------------------------------- common/minimal_job_comparison.go
struct MinimalJobRepresentatives {
    std::map<int64_t, PodGroupInfo*> representatives;
    // extractSortedResourceRequests :101-112: sort.Slice with the NON-strict LessEqual as "less" over GetPendingTasks(), which ranges a
    // map — the reference's own order is not defined for incomparable requests.  Canonical choice (same in the engine): the pending
    // pods in index order, stable insertion sort by the strict part of the comparator.
    static std::vector<const ResourceRequirements*> sortedRequests(PodGroupInfo* g) {
        std::vector<const ResourceRequirements*> v;
        for (auto* t : g->AllPodsByIndex()) if (t->status == Pending) v.push_back(&t->resReq);
        for (size_t i = 1; i < v.size(); i++) {
            const ResourceRequirements* x = v[i]; size_t k = i;
            while (k > 0 && reqLessEqual(*x, *v[k - 1]) && !reqLessEqual(*v[k - 1], *x)) { v[k] = v[k - 1]; k--; }
            v[k] = x;
        }
        return v;
    }
    static bool reqLessEqual(const ResourceRequirements& a, const ResourceRequirements& b) {  // resource_requirment.go:106-124
        if (!a.BaseResource::LessEqual(b)) return false;
        return a.GPUs() <= b.GPUs();
    }
    static bool easier(PodGroupInfo* g1, PodGroupInfo* g2) {  // jobEasierToScheduleComparison :46-76
        auto r1 = sortedRequests(g1), r2 = sortedRequests(g2);
        if (r1.empty() || r2.empty()) return false;
        if (r2.size() > r1.size()) return true;
        for (size_t i = 0; i < r1.size(); i++) {
            if (i >= r2.size()) return false;
            if (reqLessEqual(*r1[i], *r2[i])) { if (reqLessEqual(*r2[i], *r1[i])) continue; return true; }
        }
        return false;
    }
    static bool footprintSmaller(PodGroupInfo* g1, PodGroupInfo* g2) {  // isPodGroupFootprintSmaller :78-99
        auto r1 = sortedRequests(g1), r2 = sortedRequests(g2);
        if (r1.empty() || r2.empty()) return false;
        if (r1.size() > r2.size()) return false;
        for (size_t i = 0; i < r1.size(); i++) if (!reqLessEqual(*r1[i], *r2[i])) return false;
        return true;
    }
    bool IsEasierToSchedule(PodGroupInfo* other) { auto it = representatives.find(other->signature); if (it == representatives.end()) return true; return easier(other, it->second); }
    void UpdateRepresentative(PodGroupInfo* job) { auto it = representatives.find(job->signature); if (it != representatives.end() && !footprintSmaller(job, it->second)) return; representatives[job->signature] = job; }
};

// ---------------------------------------------------------------- plugins/proportion/reclaimable/{reclaimable,strategies/strategies}.go
// Reclaimable.CanReclaimResources (reclaimable.go:29-54) on one queue's attributes: what the reference's reclaimable_test.go drives directly
inline bool canReclaimResourcesCore(const QueueAttributes& q, const ResourceQuantities& requested, bool isPreemptible) {
    ResourceQuantities allocated = q.GetAllocatedShare(); for (int r = 0; r < 3; r++) allocated[r] += requested[r];
    if (!rqLessEqual(allocated, q.GetFairShare())) return false;
    if (isPreemptible) return true;
    ResourceQuantities np = q.get(&ResourceShare::AllocatedNotPreemptible); for (int r = 0; r < 3; r++) np[r] += requested[r];
    return rqLessEqual(np, q.GetDeservedShare());
}
inline bool Session::CanReclaimResources(PodGroupInfo* reclaimer) {  // reclaimable.go:29-54 via proportion.go:138-141
    if (!(cfg.plugins & KAI_PLUGIN_PROPORTION)) return false;       // session_plugins.go:117-123: no registered fn → false
    return canReclaimResourcesCore(qattrs[reclaimer->queue], QuantifyResource(GetTasksToAllocateInitResource(reclaimer, false)), reclaimer->IsPreemptibleJob());
}
struct Involved { bool r[3] = {false, false, false}; void add(const Resource& x) { if (x.milliCpu > 0) r[0] = true; if (x.memory > 0) r[1] = true; if (x.gpus > 0) r[2] = true; } void merge(const Involved& o) { for (int i = 0; i < 3; i++) r[i] = r[i] || o.r[i]; } };
inline bool reclaimableCore(const std::vector<QueueAttributes>& Q, int reclaimerQueue, const Resource& required, bool reclaimerPreemptible,
                            const std::map<int, std::vector<Resource>>& totalVictimsResources, double saturationMultiplier);
inline bool Session::reclaimableFn(Scenario* sc) {  // proportion.go:143-220 (the victims' resources by queue), then reclaimable.go:56-232
    const std::vector<QueueAttributes>& Q = jobSimulationQueues;
    PodGroupInfo* reclaimer = sc->preemptor;
    Resource required = GetTasksToAllocateInitResource(reclaimer, false);
    const bool ignoreReallocated = cfg.allow_consolidating_reclaim != 0;
    auto getResources = [&](const std::vector<PodInfo*>& pods, Resource& out) {  // proportion.go:222-240
        int n = 0; out = Resource();
        for (auto* t : pods) { if (ignoreReallocated && IsActiveAllocatedStatus(t->status)) continue; out.Add(t->accepted.AsResource()); n++; }
        return n > 0;
    };
    std::map<int, std::vector<Resource>> totalVictimsResources;  // by queue
    for (auto& kv : sc->victims) {  // scenario.GetVictims(): the tasks re-resolved by UID are the live objects here
        const VictimInfo& victim = kv.second;
        // splitVictimTasks :187-220 (sub-groups in name-rank order)
        std::vector<PodInfo*> core, elastic;
#ifdef ORC_TRACE
        ORC_T("[orc] victim job %d tasks:", victim.Job->idx); for (auto* t : victim.Tasks) ORC_T(" %d", t->idx); ORC_T("\n");
#endif
        for (auto* ps : victim.Job->podSets) {
            std::vector<PodInfo*> sub; for (auto* t : victim.Tasks) if (t->podset == ps->idx) sub.push_back(t);
            if (sub.empty()) continue;
            if (ps->minAvailable >= int32_t(sub.size())) { core.insert(core.end(), sub.begin(), sub.end()); continue; }
            core.insert(core.end(), sub.begin(), sub.begin() + ps->minAvailable); elastic.insert(elastic.end(), sub.begin() + ps->minAvailable, sub.end());
        }
        std::vector<Resource> res; Resource x;
        for (auto* t : elastic) if (getResources({t}, x)) res.push_back(x);
        if (getResources(core, x)) res.push_back(x);
        if (res.empty()) continue;
        auto& dst = totalVictimsResources[victim.Job->queue]; dst.insert(dst.end(), res.begin(), res.end());
    }
#ifdef ORC_TRACE
    for (auto& kv : totalVictimsResources) for (auto& r : kv.second) ORC_T("[orc] ent q%d %g %g %g\n", kv.first, r.milliCpu, r.memory, r.gpus);
#endif
    return reclaimableCore(Q, reclaimer->queue, required, reclaimer->IsPreemptibleJob(), totalVictimsResources, cfg.reclaimer_saturation_multiplier);
}
// reclaimable/strategies/strategies.go: MaintainFairShareStrategy :45-60 — the reclaimee is over what it may hold; GuaranteeDeservedQuotaStrategy :62-91 — the
// reclaimer stays within its deserved quota with the job and the reclaimee is over its own
inline bool maintainFairShareStrategy(const QueueAttributes& reclaimee, const ResourceQuantities& reclaimeeRemainingShare) {
    return !rqLessEqual(reclaimeeRemainingShare, reclaimee.GetAllocatableShare());
}
inline bool guaranteeDeservedQuotaStrategy(const Resource& reclaimerResources, const QueueAttributes& reclaimer, const QueueAttributes& reclaimee, const ResourceQuantities& reclaimeeRemainingShare) {
    ResourceQuantities want = reclaimer.GetAllocatedShare(), rr = QuantifyResource(reclaimerResources); for (int r = 0; r < 3; r++) want[r] += rr[r];
    if (!rqLessEqual(want, reclaimer.GetDeservedShare())) return false;  // reclaimerWillGoOverQuota :93-98
    return !rqLessEqual(reclaimeeRemainingShare, reclaimee.GetDeservedShare());
}
// Reclaimable.Reclaimable (reclaimable.go:56-232) on queue attributes, the reclaimer's queue / required resources and the reclaimees' resources by queue — the
// signature the reference's reclaimable_test.go drives
inline bool reclaimableCore(const std::vector<QueueAttributes>& Q, int reclaimerQueue, const Resource& required, bool reclaimerPreemptible,
                            const std::map<int, std::vector<Resource>>& totalVictimsResources, double saturationMultiplier) {
    auto path = [&](int q) { std::vector<int> p; for (; q >= 0; q = Q[q].parent) p.insert(p.begin(), q); return p; };  // getHierarchyPath :253-262
    // reclaimResourcesFromReclaimees :69-108
    std::map<int, ResourceQuantities> remaining; std::map<int, Involved> involved;
    for (auto& kv : totalVictimsResources) {
        int reclaimeeQueueID = kv.first;
        std::vector<int> a = path(reclaimerQueue), b = path(reclaimeeQueueID);  // getLeveledQueues :234-251
        int rq = -1, eq = -1; for (size_t i = 0; i < std::min(a.size(), b.size()); i++) { rq = a[i]; eq = b[i]; if (rq != eq) break; }
        Involved inv; for (auto& r : kv.second) inv.add(r);
        involved[reclaimeeQueueID] = inv;
        if (!remaining.count(eq)) remaining[eq] = Q[eq].GetAllocatedShare();
        for (auto& res : kv.second) {
            ResourceQuantities& rem = remaining[eq];
            // strategies.FitsReclaimStrategy :20-35
            if (!maintainFairShareStrategy(Q[eq], rem) && !guaranteeDeservedQuotaStrategy(required, Q[rq], Q[eq], rem)) return false;
            // subtractReclaimedResources :110-133
            for (int q = reclaimeeQueueID; q >= 0; q = Q[q].parent) {
                if (!remaining.count(q)) remaining[q] = Q[q].GetAllocatedShare();
                ResourceQuantities act = QuantifyResource(res); for (int r = 0; r < 3; r++) remaining[q][r] -= act[r];
                if (involved.count(q)) involved[q].merge(involved[reclaimeeQueueID]); else involved[q] = involved[reclaimeeQueueID];
            }
        }
    }
    // reclaimingQueuesRemainWithinBoundaries :135-190
    ResourceQuantities requestedQuota = QuantifyResource(required); Involved reclaimerInvolved; reclaimerInvolved.add(required);
    for (int rq = reclaimerQueue; rq >= 0; rq = Q[rq].parent) {
        ResourceQuantities rem = remaining.count(rq) ? remaining[rq] : Q[rq].GetAllocatedShare();
        for (int r = 0; r < 3; r++) rem[r] += requestedQuota[r];
        if (remaining.count(rq)) remaining[rq] = rem;  // the map holds the slice-backed value: Add mutates the stored entry
        for (auto& kv : remaining) {
            int sib = kv.first;
            if (Q[sib].parent != Q[rq].parent || sib == rq) continue;
            Involved inv = involved[sib]; inv.merge(reclaimerInvolved);
            ResourceQuantities rf = Q[rq].GetFairShare(), sf = Q[sib].GetFairShare();
            for (int r = 0; r < 3; r++) {  // isFairShareSaturationLowerPerResource :197-217
                if (!inv.r[r]) continue;
                if (rf[r] == KAI_UNLIMITED && sf[r] == KAI_UNLIMITED) continue;
                auto ratio = [](double allocated, double fair) { if (fair == 0) return allocated > 0 ? INFINITY : 0.0; if (fair == KAI_UNLIMITED) return 0.0; return allocated / fair; };
                double ratioReclaimer = ratio(rem[r], rf[r]), ratioSibling = ratio(kv.second[r], sf[r]);
                if (ratioReclaimer > 1 && sf[r] > 0 && ratioReclaimer * saturationMultiplier >= ratioSibling) return false;
            }
        }
        if (reclaimerPreemptible) continue;
        ResourceQuantities np = Q[rq].get(&ResourceShare::AllocatedNotPreemptible); for (int r = 0; r < 3; r++) np[r] += requestedQuota[r];
        if (!rqLessEqual(np, Q[rq].GetDeservedShare())) return false;
    }
    return true;
}

// ---------------------------------------------------------------- plugins/minruntime/{minruntime,resolver}.go
inline int64_t Session::preemptMinRuntime(int queue) const {  // resolvePreemptMinRuntime resolver.go:47-69: first value set from the leaf up
    for (int q = queue; q >= 0; q = queues[q].parent) if (queues[q].preemptMinRuntimeNs >= 0) return queues[q].preemptMinRuntimeNs;
    return cfg.default_preempt_min_runtime_ns;
}
inline int64_t Session::reclaimMinRuntime(int preemptorQueue, int preempteeQueue) const {
    if (cfg.reclaim_resolve_method == 1) {  // resolveReclaimMinRuntimeQueue :96-117
        for (int q = preempteeQueue; q >= 0; q = queues[q].parent) if (queues[q].reclaimMinRuntimeNs >= 0) return queues[q].reclaimMinRuntimeNs;
        return cfg.default_reclaim_min_runtime_ns;
    }
    // resolveReclaimMinRuntimeLCA :119-190
    auto path = [&](int q) { std::vector<int> p; for (; q >= 0; q = queues[q].parent) p.insert(p.begin(), q); return p; };
    std::vector<int> a = path(preemptorQueue), b = path(preempteeQueue);
    if (a[0] != b[0]) return queues[b[0]].reclaimMinRuntimeNs >= 0 ? queues[b[0]].reclaimMinRuntimeNs : cfg.default_reclaim_min_runtime_ns;
    int lca = 0; for (size_t i = 0; i < std::min(a.size(), b.size()); i++) { if (a[i] != b[i]) break; lca = int(i); }
    if (lca + 1 < int(b.size())) lca++;
    for (int i = lca; i >= 0; i--) if (queues[b[i]].reclaimMinRuntimeNs >= 0) return queues[b[i]].reclaimMinRuntimeNs;
    return cfg.default_reclaim_min_runtime_ns;
}
static inline bool jobIsElastic(const PodGroupInfo* j) { for (auto* ps : j->podSets) if (ps->IsElastic()) return true; return false; }  // job_info.go:408-415
// reclaimFilterFn / preemptFilterFn (minruntime.go:95-110): a victim still inside its min-runtime is off limits — unless it is elastic: those pass here and are checked per
// scenario by the validator below
inline bool Session::minruntimeVictimFilter(const PodGroupInfo* pending, const PodGroupInfo* victim, bool reclaim) const {
    if (!minruntimeOn() || jobIsElastic(victim)) return true;
    return !isProtected(victim, reclaim ? reclaimMinRuntime(pending->queue, victim->queue) : preemptMinRuntime(victim->queue));
}
inline bool Session::minruntimeValidator(Scenario* sc, bool reclaim) {  // reclaimScenarioValidatorFn / preemptScenarioValidatorFn minruntime.go:112-142
    for (auto& kv : sc->victims) {
        const VictimInfo& v = kv.second;
        if (!jobIsElastic(v.Job)) continue;
        int64_t mr = reclaim ? reclaimMinRuntime(sc->preemptor->queue, v.Job->queue) : preemptMinRuntime(v.Job->queue);
        if (!isProtected(v.Job, mr)) continue;
        // validVictimForMinAvailable :198-222: the protected elastic job must keep minAvailable tasks in every touched sub-group
        for (auto* ps : v.Job->podSets) {
            int numVictims = 0; for (auto* t : v.Tasks) if (t->podset == ps->idx) numVictims++;
            if (!numVictims) continue;
            if (ps->minAvailable > int32_t(ps->numActiveUsedTasks) - int32_t(numVictims)) return false;
        }
    }
    return true;
}

// ---------------------------------------------------------------- actions/utils/action.go:18-47
inline std::unique_ptr<JobsOrderByQueues> Session::GetVictimsQueue(const std::function<bool(PodGroupInfo*)>& filter) {
    std::vector<PodGroupInfo*> preemptees;
    for (auto& job : jobs) {
        bool alive = false; for (auto* t : job.AllPods()) if (IsAliveStatus(t->status)) alive = true;
        if (alive && (!filter || filter(&job))) preemptees.push_back(&job);
    }
    JobsOrderInitOptions o; o.VictimQueue = true; o.MaxJobsQueueDepth = -1;
    auto q = std::make_unique<JobsOrderByQueues>(this, o);
    q->InitializeWithJobs(preemptees);
    return q;
}
static inline int activeAllocatedCount(PodGroupInfo* job) { int n = 0; for (auto* t : job->AllPods()) if (IsActiveAllocatedStatus(t->status)) n++; return n; }

// ---------------------------------------------------------------- actions/{reclaim,preempt,consolidation}
inline void Session::executeVictimAction(int action) {
    JobsOrderInitOptions o; o.FilterNonPending = true; o.FilterUnready = true;
    if (action == KAI_ACTION_CONSOLIDATION) { if (cfg.max_consolidation_preemptees == 0) return; o.FilterNonPreemptible = true; }  // consolidation.go:36-46
    o.MaxJobsQueueDepth = cfg.queue_depth[action] == 0 ? -1 : cfg.queue_depth[action];
    JobsOrderByQueues jobsOrder(this, o);
    std::vector<PodGroupInfo*> all; for (auto& j : jobs) all.push_back(&j);
    jobsOrder.InitializeWithJobs(all);
    std::map<int, MinimalJobRepresentatives> smallestFailedJobsByQueue;  // consolidation keeps ONE set (consolidation.go:52): key -1
    while (!jobsOrder.IsEmpty()) {
        PodGroupInfo* job = jobsOrder.PopNextJob(); if (!job) break;
        if (action == KAI_ACTION_RECLAIM && !CanReclaimResources(job)) continue;  // reclaim.go:64-66
        MinimalJobRepresentatives& smallest = smallestFailedJobsByQueue[action == KAI_ACTION_CONSOLIDATION ? -1 : job->queue];
        if (cfg.use_scheduling_signatures && !smallest.IsEasierToSchedule(job)) { ORC_T("[orc] mjr skip job %d\n", job->idx); continue; }
        ORC_T("[orc] attempt job %d\n", job->idx);
        stats.jobsAttempted++;
        std::unique_ptr<Statement> stmt; bool ok = false;
        clonePool.clear();
        switch (action) {
            case KAI_ACTION_RECLAIM: {  // reclaim.go:102-143
                GetTasksToAllocateInitResource(job, false);
                jobSimulationQueues = qattrs;  // ssn.OnJobSolutionStart → proportion.OnJobSolutionStartFn (proportion.go:131-136)
                JobSolver solver{this, FeasibleNodesForJob(job), [this](Scenario* sc) { if ((cfg.plugins & KAI_PLUGIN_PROPORTION) && !reclaimableFn(sc)) return false; return !minruntimeOn() || minruntimeValidator(sc, true); },
                    [this, job]() {
                        JobsOrderInitOptions vo; vo.FilterNonPreemptible = true; vo.FilterNonActiveAllocated = true; vo.VictimQueue = true; vo.MaxJobsQueueDepth = -1;
                        auto q = std::make_unique<JobsOrderByQueues>(this, vo);
                        std::vector<PodGroupInfo*> v;
                        for (auto& other : jobs) {
                            if (other.queue == job->queue) continue;
                            // ssn.ReclaimVictimFilter → minruntime.reclaimFilterFn (minruntime.go:95-101): elastic jobs pass, they are checked per scenario
                            if (!minruntimeVictimFilter(job, &other, true)) continue;
                            v.push_back(&other);
                        }
                        q->InitializeWithJobs(v); return q;
                    }};
                ok = solver.Solve(job, stmt);
                break;
            }
            case KAI_ACTION_PREEMPT: {  // preempt.go:99-161
                GetTasksToAllocateInitResource(job, false);
                std::vector<PodInfo*> preemptorTasks = GetTasksToAllocate(job, false);
                if (cfg.plugins & KAI_PLUGIN_PROPORTION) {  // IsNonPreemptibleJobOverQueueQuotaFn → capacity_policy.go:38-49
                    ResourceQuantities q{0, 0, 0}; for (auto* pod : preemptorTasks) { q[2] += pod->resReq.GetGpusQuota(); q[0] += pod->resReq.milliCpu; q[1] += pod->resReq.memory; }
                    if (resultsWithNonPreemptibleOverQuota(q, job)) { ORC_T("[orc] np over quota job %d req %g %g %g ntasks %d\n", job->idx, q[0], q[1], q[2], (int)preemptorTasks.size()); break; }
                }
                JobSolver solver{this, FeasibleNodesForJob(job), [this](Scenario* sc) { return !minruntimeOn() || minruntimeValidator(sc, false); },
                    [this, job]() { return GetVictimsQueue([this, job](PodGroupInfo* v) {  // buildFilterFuncForPreempt :122-152
                        if (!v->IsPreemptibleJob()) return false;
                        if (v->priority >= job->priority) return false;
                        if (v->queue != job->queue) return false;
                        if (v->idx == job->idx) return false;
                        if (activeAllocatedCount(v) == 0) return false;
                        if (!minruntimeVictimFilter(job, v, false)) return false;  // PreemptVictimFilter → minruntime.preemptFilterFn :103-110
                        return true; }); }};
                ok = solver.Solve(job, stmt);
                break;
            }
            case KAI_ACTION_CONSOLIDATION: {  // consolidation.go:80-157
                GetTasksToAllocateInitResource(job, false);
                double sumGpus = 0;  // utils.IsEnoughGPUsAllocatableForJob (action.go:119-160)
                for (auto& n : nodes) { if (n.flags & KAI_NODE_NOT_READY) continue; sumGpus += n.GetSumOfIdleGPUs(); sumGpus += n.GetSumOfReleasingGPUs(); }
                int64_t sumMem = 0; for (auto& n : nodes) { if (n.flags & KAI_NODE_NOT_READY) continue; sumMem += n.GetSumOfIdleGPUsMemory(); sumMem += n.GetSumOfReleasingGPUsMemory(); }
                double requested = 0; int64_t requestedMem = 0;  // GetTasksToAllocateRequestedGPUs (allocation_info.go:55-86)
                for (auto* t : GetTasksToAllocate(job, false)) {
                    requested += t->resReq.GPUs(); requestedMem += t->resReq.gpuMemory;
                    if (migRows().any) for (auto& kv : t->resReq.scalars) if (kv.first < KAI_MAX_RES && migRows().gpus[kv.first] > 0) { requested += double(int64_t(migRows().gpus[kv.first]) * kv.second); requestedMem += migRows().mem[kv.first] * kv.second; }  // :73-81
                }
                if (!(sumGpus >= requested && sumMem >= requestedMem)) break;
                JobSolver solver{this, FeasibleNodesForJob(job),
                    [](Scenario* sc) { for (auto& kv : sc->victims) for (auto* t : kv.second.Tasks) if (t->status == Releasing) return false; return true; },  // allPodsReallocated :120-129
                    [this, job]() {
                        auto counter = std::make_shared<int>(0); int maxPreemptees = cfg.max_consolidation_preemptees;
                        return GetVictimsQueue([job, counter, maxPreemptees](PodGroupInfo* v) {  // buildPreemptibleFilterFunc :136-157
                            if (!v->IsPreemptibleJob()) return false;
                            if (v->idx == job->idx) return false;
                            if (maxPreemptees != -1 && *counter > maxPreemptees) return false;
                            if (activeAllocatedCount(v) == 0) return false;
                            *counter += 1; return true; }); }};
                ok = solver.Solve(job, stmt);
                break;
            }
        }
        if (ok && stmt) { stats.jobsCommitted++; stmt->Commit(); }
        else smallest.UpdateRepresentative(job);
    }
    clonePool.clear();
}

}  // namespace orc
