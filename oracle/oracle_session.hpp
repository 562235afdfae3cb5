// oracle_session.hpp — TEST INFRASTRUCTURE (see oracle_model.hpp header).
// Restates framework.Session / Statement and the default plugin chain of the reference for the hot path.
#pragma once
#include <algorithm>
#include <array>
#include <cfloat>
#include <cstdio>

#include "oracle_model.hpp"

namespace orc {

struct Session; struct Scenario; struct JobClone; struct JobsOrderByQueues;

// ---------------------------------------------------------------- framework/statement.go + operations.go
enum OpName { opEvict, opPipeline, opAllocate, opUndo };
struct Operation {
    OpName name; PodInfo* task = nullptr;
    int previousStatus = 0; int previousNode = -1; int nextNode = -1; bool previousIsVirtual = false;
    int operationIndex = -1;  // undo: index of the undone op
    std::vector<int> previousGpuGroups;  // evictOperation / pipelineOperation.previousGpuGroups (operations.go)
};

struct Statement {
    Session* ssn; std::vector<Operation> operations;
    explicit Statement(Session* s) : ssn(s) {}
    int Checkpoint() const { return int(operations.size()); }            // statement.go:44-46
    void Rollback(int cp) {                                                // statement.go:48-61
        for (int i = int(operations.size()) - 1; i >= cp; i--) undoOperation(i);
        operations.resize(cp);
    }
    bool Evict(PodInfo* task);                                             // statement.go:63-126
    bool unevict(PodInfo* task, int previousStatus, int node, bool previousIsVirtual, const std::vector<int>& previousGpuGroups);  // :152-195
    bool Pipeline(PodInfo* task, int node, bool updateTaskIfExistsOnNode); // :197-295
    bool Allocate(PodInfo* task, int node);                                // :297-358
    bool unallocate(PodInfo* task, bool previousIsVirtual);                // :391-425
    bool unpipeline(PodInfo* task, int previousNode, int previousStatus, bool previousIsVirtual, const std::vector<int>& previousGpuGroups);  // :431-476
    bool Unevict(PodInfo* task) { return undoEarliestValidOperation(task, opEvict); }              // :478-481
    bool ConvertAllAllocatedToPipelined(int jobIdx);                       // :483-516
    void Discard() { for (int i = int(operations.size()) - 1; i >= 0; i--) undoOperation(i); operations.clear(); }  // :522-534
    void Commit();                                                         // :536-575
    bool undoEarliestValidOperation(PodInfo* task, OpName name) {          // :578-600
        for (int i = 0; i < int(operations.size()); i++) {
            if (!operationValid(i)) continue;
            if (operations[i].task != task || operations[i].name != name) continue;
            undoOperation(i); return true;
        }
        return false;
    }
    void undoOperation(int index);                                         // :602-643
    bool operationValid(int i) const {                                     // :652-663
        for (int u = 0; u < int(operations.size()); u++) {
            if (operations[u].name != opUndo) continue;
            if (operations[u].operationIndex == i) return !operationValid(u);
        }
        return true;
    }
};

struct SessionStats { int64_t decisions = 0, nodeScans = 0, nodesScanned = 0, jobsAttempted = 0, jobsCommitted = 0, rollbacks = 0, scenarios = 0, simulations = 0, scenariosFiltered = 0; };

// ---------------------------------------------------------------- framework/session.go:51-90 + the default tier
struct Session {
    kai_config cfg{};
    int R = 4;
    std::vector<NodeInfo> nodes; std::vector<PodInfo> pods; std::vector<PodSet> podsets; std::vector<PodGroupInfo> jobs; std::vector<QueueInfo> queues;
    std::vector<SubGroupSet> groups;
    // topology plugin state (plugins/topology/topology_plugin.go:22-28): domain trees + per-sub-group node scores of the current job
    int nTopologies = 0; std::vector<int> topoLevelOff, nodeDomain; std::vector<DomainInfo> domains;  // domains[D + t] = root domain of topology t
    int nRealDomains = 0;
    std::map<int, std::map<int, double>> subGroupNodeScores;  // key: group idx, or -(podset idx + 1)
    int lastCommonDomain = -2; std::vector<int> lastValidNodes;  // what lowestCommonDomainID returned in the last SubsetNodesFn (read by kai_oracle_lowest_common_domain: common_test.go)
    std::vector<uint8_t> classFit; int nPodClasses = 0, nNodeClasses = 0;
    bool hasSignatures = false;  // the snapshot carries job_signature
    // proportion plugin state (plugins/proportion/proportion.go:52-65)
    ResourceQuantities totalResource{0, 0, 0};
    std::vector<QueueAttributes> qattrs;
    // nodeplacement plugin state (plugins/nodeplacement/nodeplacement.go:29-31)
    std::map<int, std::pair<double, double>> podAllocatableRange;
    // committed operations in commit order (what cache.Bind / Evict / TaskPipelined would receive)
    std::vector<kai_op> committed;
    int32_t n_statements = 0;  // Statements that committed at least one operation so far (kai_op.stmt)
    SessionStats stats;

    void load(const kai_config* c, const kai_snapshot_soa* s);

    // ---- plugins/proportion
    void proportionOnSessionOpen();
    void allocateHandler(PodInfo* task);    // proportion.go:443-465
    void deallocateHandler(PodInfo* task);  // proportion.go:467-489
    int queueOrder(int lQ, int rQ, PodGroupInfo* lJob, PodGroupInfo* rJob, const std::vector<PodGroupInfo*>& lVictims, const std::vector<PodGroupInfo*>& rVictims);
    bool IsJobOverQueueCapacity(PodGroupInfo* job, const std::vector<PodInfo*>& tasks);            // capacity_policy.go:26-36
    bool IsTaskAllocationOnNodeOverCapacity(PodInfo* task, PodGroupInfo* job, NodeInfo* node);     // capacity_policy.go:51-61
    bool resultsOverLimit(const ResourceQuantities& req, PodGroupInfo* job);
    bool resultsWithNonPreemptibleOverQuota(const ResourceQuantities& req, PodGroupInfo* job);

    // ---- order fns (framework/session_plugins.go:227-299)
    bool JobOrderFn(PodGroupInfo* l, PodGroupInfo* r);
    bool TaskOrderFn(PodInfo* l, PodInfo* r);
    bool PodSetOrderFn(PodSet* l, PodSet* r);
    bool QueueOrderFn(int lQ, int rQ, PodGroupInfo* lJob, PodGroupInfo* rJob, const std::vector<PodGroupInfo*>& lV, const std::vector<PodGroupInfo*>& rV);

    // ---- podgroup_info/allocation_info.go
    const std::vector<PodInfo*>& GetTasksToAllocate(PodGroupInfo* job, bool isRealAllocation);
    const Resource& GetTasksToAllocateInitResource(PodGroupInfo* job, bool isRealAllocation);
    bool HasTasksToAllocate(PodGroupInfo* job, bool isRealAllocation) { for (auto* t : job->AllPods()) if (t->ShouldAllocate(isRealAllocation)) return true; return false; }

    // ---- node ordering / predicates (framework/session.go:201-283, session_plugins.go:393-437)
    void NodePreOrderFn(PodInfo* task, const std::vector<NodeInfo*>& nodes);
    double NodeOrderFn(PodInfo* task, NodeInfo* node);
    std::vector<NodeInfo*> OrderedNodesByTask(const std::vector<NodeInfo*>& nodes, PodInfo* task);
    bool FittingNode(PodInfo* task, NodeInfo* node);
    bool PredicateFn(PodInfo* task, PodGroupInfo* job, NodeInfo* node);

    // ---- actions/common/allocate.go
    bool AllocateJob(Statement& stmt, const std::vector<NodeInfo*>& nodes, PodGroupInfo* job, bool isPipelineOnly);
    bool allocateSubGroupSet(Statement& stmt, const std::vector<NodeInfo*>& nodes, PodGroupInfo* job, SubGroupSet* sgs, const std::vector<PodInfo*>& tasks, bool isPipelineOnly);
    bool allocateSubGroupSetOnNodes(Statement& stmt, const std::vector<NodeInfo*>& nodes, PodGroupInfo* job, SubGroupSet* sgs, const std::vector<PodInfo*>& tasks, bool isPipelineOnly);
    bool allocatePodSet(Statement& stmt, const std::vector<NodeInfo*>& nodes, PodGroupInfo* job, PodSet* ps, const std::vector<PodInfo*>& tasks, bool isPipelineOnly);
    // ---- plugins/topology
    void allPodSets(PodGroupInfo* job, SubGroupSet* sgs, std::vector<PodSet*>& out);
    bool SubsetNodesFn(PodGroupInfo* job, int key, const TopologyConstraint& tc, const std::vector<PodSet*>& podSets, const std::vector<PodInfo*>& tasks,
                       const std::vector<NodeInfo*>& nodeSet, std::vector<std::vector<NodeInfo*>>& out);
    double topologyNodeScore(PodInfo* task, NodeInfo* node, bool& err);
    bool allocateTask(Statement& stmt, const std::vector<NodeInfo*>& nodes, PodInfo* task, bool isPipelineOnly);
    bool allocateTaskToNode(Statement& stmt, PodInfo* task, NodeInfo* node, bool isPipelineOnly);
    // shared GPUs: framework/session.go:163-199, gpu_sharing/gpuSharing.go, plugins/{gpupack,gpuspread,gpusharingorder}
    struct NodeGpuForSharing { std::vector<int> Groups; bool IsReleasing = false; bool ok = false; };
    int nextNewGpuGroup = kNewGpuGroup;  // uuid.NewUUID() of findGpuForSharingOnNode
    bool hasFractions = false;
    double GpuOrderFn(PodInfo* task, NodeInfo* node, int gpu, bool* err = nullptr);
    std::vector<int> FittingGPUs(NodeInfo* node, PodInfo* pod);
    NodeGpuForSharing GetNodePreferableGpuForSharing(const std::vector<int>& fittingGPUs, NodeInfo* node, PodInfo* pod, bool isPipelineOnly);
    bool AllocateFractionalGPUTaskToNode(Statement& stmt, PodInfo* pod, NodeInfo* node, bool isPipelineOnly);
    bool willCreateNewGpuGroup(PodInfo* task, NodeInfo* node);

    // ---- actions
    void executeAllocate();  // actions/allocate/allocate.go:46-77
    // ---- victim search (oracle_solver.hpp): actions/{reclaim,preempt,consolidation} + actions/common/solvers
    std::vector<std::unique_ptr<JobClone>> clonePool;  // CloneWithTasks representatives of the job being solved
    std::vector<QueueAttributes> jobSimulationQueues;  // proportion.OnJobSolutionStartFn (proportion.go:131-136)
    PodGroupInfo* CloneWithTasks(PodGroupInfo* src, const std::vector<PodInfo*>& tasks);
    std::vector<PodInfo*> GetTasksToEvict(PodGroupInfo* job, bool& hasMoreTasks);
    std::vector<int> FeasibleNodesForJob(PodGroupInfo* job);
    bool CanReclaimResources(PodGroupInfo* reclaimer);
    bool reclaimableFn(Scenario* sc);
    std::unique_ptr<JobsOrderByQueues> GetVictimsQueue(const std::function<bool(PodGroupInfo*)>& filter);
    void executeVictimAction(int action);
    // ---- plugins/minruntime
    bool minruntimeOn() const { return (cfg.plugins & KAI_PLUGIN_MINRUNTIME) != 0; }
    int64_t preemptMinRuntime(int queue) const;
    int64_t reclaimMinRuntime(int preemptorQueue, int preempteeQueue) const;
    bool isProtected(const PodGroupInfo* victim, int64_t minRuntime) const { return victim->lastStartNs != 0 && cfg.now_ns < victim->lastStartNs + minRuntime; }
    bool minruntimeValidator(Scenario* sc, bool reclaim);
    bool minruntimeVictimFilter(const PodGroupInfo* pending, const PodGroupInfo* victim, bool reclaim) const;  // reclaimFilterFn / preemptFilterFn: true = the victim may be taken
    ~Session();
};

// ---------------------------------------------------------------- actions/utils/job_order_by_queue.go
struct JobsOrderInitOptions { bool FilterNonPending = false, FilterUnready = false, FilterNonPreemptible = false, FilterNonActiveAllocated = false, VictimQueue = false; int MaxJobsQueueDepth = -1; };

struct queueNode {
    int queue = -1; bool isLeaf = false; bool needsReorder = false; queueNode* parent = nullptr;
    PriorityQueue<queueNode*> childNodes;     // non-leaf
    PriorityQueue<PodGroupInfo*> childJobs;   // leaf
    bool childrenEmpty() const { return isLeaf ? childJobs.Empty() : childNodes.Empty(); }
    int childrenLen() const { return isLeaf ? childJobs.Len() : childNodes.Len(); }
};

struct JobsOrderByQueues {
    Session* ssn; JobsOrderInitOptions options;
    bool rootInit = false; PriorityQueue<queueNode*> rootNodes;
    std::map<int, std::unique_ptr<queueNode>> queueNodes;
    std::vector<std::unique_ptr<queueNode>> graveyard;  // nodes deleted from the map while still referenced by parents' pointers
    std::map<int, std::vector<PodGroupInfo*>> poppedJobsByQueue;
    JobsOrderByQueues(Session* s, JobsOrderInitOptions o) : ssn(s), options(o) {}
    bool IsEmpty() const { return !rootInit || rootNodes.Empty(); }
    int Len() const { int c = 0; for (auto& kv : queueNodes) if (kv.second->isLeaf) c += kv.second->childJobs.Len(); return c; }
    void InitializeWithJobs(const std::vector<PodGroupInfo*>& jobsToOrder);  // input_jobs.go:21-68
    PodGroupInfo* PopNextJob();
    void PushJob(PodGroupInfo* job);
    std::function<bool(queueNode* const&, queueNode* const&)> buildNodeOrderFn(bool reverseOrder);
    std::pair<PodGroupInfo*, std::vector<PodGroupInfo*>> getBestJobFromNode(queueNode* node);
    queueNode* getNextNode(PriorityQueue<queueNode*>& pq);
    queueNode* traverseToLeaf(PriorityQueue<queueNode*>& pq);
    void handlePopFromNode(queueNode* node);
    void markAncestorsForReorder(queueNode* node) { for (queueNode* c = node; c; c = c->parent) c->needsReorder = true; }
    void ensureAncestorChainForPush(queueNode* childNode, const QueueInfo& childQueue);
};

}  // namespace orc
