// Package gpucore routes the placement actions of the KAI scheduler (allocate, consolidation, reclaim, preempt) through
// libkai_core on an MI355X.  It is the reference-side binding of include/kai_core.h: a framework.Plugin that packs the
// session into the C ABI's structure-of-arrays at OnSessionOpen, and framework.Actions that run kai_action_execute and
// replay the committed operations through the real Statement, so that cache.Bind / Evict, fit errors, status updates
// and metrics stay on the Go side.
//
// Drop this file into pkg/scheduler/gpucore/ of NVIDIA/KAI-Scheduler (paths below are under pkg/scheduler/).  The build
// image of this repository has no Go toolchain, so the file is shipped as source, not compiled here; the host-side mirror
// that IS exercised is kai-scheduler_amd/core.py (same calls, same order) and the packing rules are pinned by
// tests/kai_testlib.py::case_to_snapshot and kai_ingest.cpp, which pack the same fields from the reference's fixtures
// and from snapshot.json.
package gpucore

/*
#cgo CFLAGS:  -I${SRCDIR}/../../../third_party/kai_core/include
#cgo LDFLAGS: -L${SRCDIR}/../../../third_party/kai_core/lib -lkai_core
#include <stdlib.h>
#include <string.h>
#include "kai_core.h"
*/
import "C"

import (
	"fmt"
	"sort"
	"unsafe"

	v1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/types"

	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/actions/allocate"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/actions/consolidation"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/actions/preempt"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/actions/reclaim"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/actions/utils"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/common_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/eviction_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/node_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/pod_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/pod_status"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/podgroup_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/queue_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/framework"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/scheduler_util"
)

const nRes = 4 // cpu (milli), memory (bytes), gpu (devices), pods — api/resource_info/resource_vector.go:23-36.  Rows >= 4 (other scalar resources; MIG instance
// types with res_mig_gpus / res_mig_memory beside them, ABI v5) are laid out as kai_ingest.cpp does from the snapshot file; this shim packs the four base rows and leaves
// tasks with MIG instances to the fallback rule below.

var core *C.kai_core // one handle per scheduler process (= per scheduling shard)

// Init creates the device handle.  cfg mirrors conf.SchedulerParams and the plugin arguments the path reads.
func Init(cfg C.kai_config, gpu int) error {
	id := C.int(gpu)
	if rc := C.kai_core_create(&cfg, 1, &id, &core); rc != 0 {
		return fmt.Errorf("kai_core_create: status %d", int(rc))
	}
	return nil
}

// ------------------------------------------------------------------------------------------------ packed snapshot
// Everything the ABI takes lives in C memory (cgo forbids retaining Go pointers); freed at OnSessionClose.
type packedSnapshot struct {
	soa      C.kai_snapshot_soa
	allocs   []unsafe.Pointer
	pods     []*pod_info.PodInfo           // ABI pod index  -> task
	nodes    []*node_info.NodeInfo         // ABI node index -> node
	jobs     []*podgroup_info.PodGroupInfo // ABI job index  -> pod group
	classes  *staticClasses                // pod / node classes of the static Filters, shared-GPU group ids
	fallback bool                          // the rest of this cycle runs on the Go actions
}

func (p *packedSnapshot) free() {
	for _, a := range p.allocs {
		C.free(a)
	}
	p.allocs = nil
}

func carray[T any](p *packedSnapshot, n int) []T {
	var zero T
	sz := C.size_t(max(n, 1)) * C.size_t(unsafe.Sizeof(zero))
	mem := C.malloc(sz)
	C.memset(mem, 0, sz)
	p.allocs = append(p.allocs, mem)
	return unsafe.Slice((*T)(mem), max(n, 1))[:n:n]
}
func ptr[T any](s []T) *T {
	if len(s) == 0 {
		return nil
	}
	return &s[0]
}

// ranks of strings in byte-wise order: the reference's tie-breaks are Go string compares
// (framework/session.go:480-485 node names; framework/session_plugins.go:227-260 job / pod UIDs)
func rankStrings(keys []string) []C.uint32_t {
	idx := make([]int, len(keys))
	for i := range idx {
		idx[i] = i
	}
	sort.SliceStable(idx, func(a, b int) bool { return keys[idx[a]] < keys[idx[b]] })
	out := make([]C.uint32_t, len(keys))
	for r, i := range idx {
		out[i] = C.uint32_t(r)
	}
	return out
}

func statusBit(s pod_status.PodStatus) C.int32_t { // api/pod_status/pod_status.go:25-71 in declaration order
	switch s {
	case pod_status.Pending:
		return C.KAI_POD_PENDING
	case pod_status.Gated:
		return C.KAI_POD_GATED
	case pod_status.Allocated:
		return C.KAI_POD_ALLOCATED
	case pod_status.Pipelined:
		return C.KAI_POD_PIPELINED
	case pod_status.Binding:
		return C.KAI_POD_BINDING
	case pod_status.Bound:
		return C.KAI_POD_BOUND
	case pod_status.Running:
		return C.KAI_POD_RUNNING
	case pod_status.Releasing:
		return C.KAI_POD_RELEASING
	case pod_status.Succeeded:
		return C.KAI_POD_SUCCEEDED
	case pod_status.Failed:
		return C.KAI_POD_FAILED
	case pod_status.Deleted:
		return C.KAI_POD_DELETED
	}
	return C.KAI_POD_UNKNOWN
}

// podModelFlags: what the device path does not carry for this task (SURVEY §8b fallback rule; kai_ingest.cpp:592-611 sets the same bits from snapshot.json).
//   KAI_POD_CPU_FALLBACK   a state-dependent upstream predicate or a request kind without a device model: a PENDING pod with it makes kai_session_open decline
//   KAI_POD_GPU_UNMODELLED several fractional devices / DRA claims: an ACTIVE pod with it holds GPU state the node accounting would overstate, the session is declined
//   KAI_POD_LEGACY_MIG     a legacy MIG task (annotation-named instances): never scheduled, and its node takes no MIG request (node_info.go:315-359, 407-409)
// A fraction or MiB of ONE device (ABI v4 / v5) and MIG instance requests (resource rows >= 4) are described to the device and need no flag.
func podModelFlags(t *pod_info.PodInfo) C.uint32_t {
	var f C.uint32_t
	if t.IsLegacyMIGtask {
		f |= C.KAI_POD_LEGACY_MIG
	}
	multiFraction := t.ResReq.GetNumOfGpuDevices() > 1 && t.ResReq.IsFractionalRequest()
	dra := t.ResReq.GetDraGpusCount() > 0 || len(t.Pod.Spec.ResourceClaims) > 0
	if multiFraction || dra {
		f |= C.KAI_POD_CPU_FALLBACK | C.KAI_POD_GPU_UNMODELLED
	}
	if len(t.ResReq.MigResources()) > 0 {
		f |= C.KAI_POD_CPU_FALLBACK // this shim packs the four base resource rows only (nRes): MIG instance rows are laid out by the snapshot-file path
	}
	spec := t.Pod.Spec
	if spec.Affinity != nil && (spec.Affinity.PodAffinity != nil || spec.Affinity.PodAntiAffinity != nil) {
		f |= C.KAI_POD_CPU_FALLBACK
	}
	for _, cs := range [][]v1.Container{spec.Containers, spec.InitContainers} {
		for _, c := range cs {
			for _, port := range c.Ports {
				if port.HostPort != 0 {
					f |= C.KAI_POD_CPU_FALLBACK
				}
			}
		}
	}
	if len(t.GetAllStorageClaims()) > 0 {
		f |= C.KAI_POD_CPU_FALLBACK
	}
	return f
}

// packSnapshot: ssn.ClusterInfo (api/cluster_info.go:43-64, built by cache/cluster_info/cluster_info.go:118-228) ->
// kai_snapshot_soa, field by field as include/kai_core.h documents them.
func packSnapshot(ssn *framework.Session, params packParams) *packedSnapshot {
	p := &packedSnapshot{}
	ci := ssn.ClusterInfo
	s := &p.soa
	s.abi_version = C.KAI_ABI_VERSION
	s.n_res = nRes

	// ---- nodes, in map-iteration-independent (name) order; the library re-permutes by node_name_rank itself
	names := make([]string, 0, len(ci.Nodes))
	for name := range ci.Nodes {
		names = append(names, name)
	}
	sort.Strings(names)
	N := len(names)
	nodeIdx := make(map[string]int, N)
	alloc := carray[C.double](p, nRes*N)
	flags := carray[C.uint32_t](p, N)
	gpuCount := carray[C.int32_t](p, N)
	nodeClass := carray[C.int32_t](p, N)
	gpuMem := carray[C.int64_t](p, N)
	for i, name := range names {
		n := ci.Nodes[name]
		nodeIdx[name] = i
		p.nodes = append(p.nodes, n)
		alloc[0*N+i] = C.double(n.Allocatable.Cpu())
		alloc[1*N+i] = C.double(n.Allocatable.Memory())
		alloc[2*N+i] = C.double(n.Allocatable.GPUs())
		alloc[3*N+i] = C.double(n.Allocatable.Get(v1.ResourcePods))
		if err := scheduler_util.CheckNodeConditionPredicate(n.Node); err != nil { // scheduler_util/scheduler_utils.go:12-40
			flags[i] |= C.KAI_NODE_NOT_READY
		}
		if n.IsMIGEnabled() { // api/node_info/node_info.go:704-718
			flags[i] |= C.KAI_NODE_MIG_ENABLED
		}
		switch n.GetMigStrategy() { // :720-732
		case node_info.MigStrategyMixed:
			flags[i] |= C.KAI_NODE_MIG_MIXED
		case node_info.MigStrategySingle:
			flags[i] |= C.KAI_NODE_MIG_SINGLE
		}
		if n.HasDRAGPUs {
			flags[i] |= C.KAI_NODE_HAS_DRA_GPUS
		}
		if _, ok := n.Node.Labels[params.gpuWorkerLabel]; ok { // plugins/predicates/predicates.go:243-259
			flags[i] |= C.KAI_NODE_GPU_WORKER
		}
		if _, ok := n.Node.Labels[params.cpuWorkerLabel]; ok {
			flags[i] |= C.KAI_NODE_CPU_WORKER
		}
		gpuCount[i] = -1
		if c, ok := n.GetLabelGpuCount(); ok { // nvidia.com/gpu.count (node_info.go:619-640)
			gpuCount[i] = C.int32_t(c)
		}
		gpuMem[i] = C.int64_t(n.MemoryOfEveryGpuOnNode)
	}
	nodeRank := rankStrings(names)
	cNodeRank := carray[C.uint32_t](p, N)
	copy(cNodeRank, nodeRank)

	// ---- queues (leaf queues and departments share one index space); parents before use is not required
	qids := make([]string, 0, len(ci.Queues))
	for id := range ci.Queues {
		qids = append(qids, string(id))
	}
	sort.Strings(qids)
	Q := len(qids)
	queueIdx := make(map[string]int, Q)
	for i, id := range qids {
		queueIdx[id] = i
	}
	qParent := carray[C.int32_t](p, Q)
	qPrio := carray[C.int32_t](p, Q)
	qCreated := carray[C.int64_t](p, Q)
	qUID := carray[C.uint32_t](p, Q)
	qDeserved := carray[C.double](p, 3*Q)
	qLimit := carray[C.double](p, 3*Q)
	qOqw := carray[C.double](p, 3*Q)
	qUsage := carray[C.double](p, 3*Q)
	qPreMR := carray[C.int64_t](p, Q)
	qRecMR := carray[C.int64_t](p, Q)
	copy(qUID, rankStrings(qids))
	for i, id := range qids {
		q := ci.Queues[queue_info.QueueID(id)] // import alias of common_info.QueueID in the reference
		qParent[i] = -1
		if pi, ok := queueIdx[string(q.ParentQueue)]; ok && q.ParentQueue != "" {
			qParent[i] = C.int32_t(pi)
		}
		qPrio[i] = C.int32_t(q.Priority)
		qCreated[i] = C.int64_t(q.CreationTimestamp.UnixNano())
		for k, rq := range []queue_info.ResourceQuota{q.Resources.CPU, q.Resources.Memory, q.Resources.GPU} { // KAI_Q_CPU, _MEM, _GPU
			qDeserved[k*Q+i] = C.double(rq.Quota) // memory stays in the CRD's 10^6-byte units: the library applies proportion.go:327-328
			qLimit[k*Q+i] = C.double(rq.Limit)
			qOqw[k*Q+i] = C.double(rq.OverQuotaWeight)
		}
		if u, ok := ci.QueueResourceUsage.Queues[q.UID]; ok { // normalised historical usage (api/queue_info/quota_info.go)
			qUsage[0*Q+i], qUsage[1*Q+i], qUsage[2*Q+i] = C.double(u[v1.ResourceCPU]), C.double(u[v1.ResourceMemory]), C.double(u["nvidia.com/gpu"])
		}
		qPreMR[i], qRecMR[i] = -1, -1
		if q.PreemptMinRuntime != nil {
			qPreMR[i] = C.int64_t(q.PreemptMinRuntime.Duration.Nanoseconds())
		}
		if q.ReclaimMinRuntime != nil {
			qRecMR[i] = C.int64_t(q.ReclaimMinRuntime.Duration.Nanoseconds())
		}
	}

	// ---- jobs, pod-sets, pods: pods of one job contiguous, pod-sets of one job contiguous
	jids := make([]string, 0, len(ci.PodGroupInfos))
	for id := range ci.PodGroupInfos {
		jids = append(jids, string(id))
	}
	sort.Strings(jids)
	J := len(jids)
	P, S := 0, 0
	for _, id := range jids {
		job := ci.PodGroupInfos[podgroup_info.PodGroupID(id)]
		P += len(job.GetAllPodsMap())
		S += len(job.PodSets)
	}
	jQueue := carray[C.int32_t](p, J)
	jPrio := carray[C.int32_t](p, J)
	jPreempt := carray[C.int32_t](p, J)
	jCreated := carray[C.int64_t](p, J)
	jUID := carray[C.uint32_t](p, J)
	jFirstPod := carray[C.int32_t](p, J)
	jNPods := carray[C.int32_t](p, J)
	jFirstPS := carray[C.int32_t](p, J)
	jNPS := carray[C.int32_t](p, J)
	jSig := carray[C.int64_t](p, J)
	jLastStart := carray[C.int64_t](p, J)
	psJob := carray[C.int32_t](p, S)
	psMin := carray[C.int32_t](p, S)
	psRank := carray[C.uint32_t](p, S)
	req := carray[C.double](p, nRes*P)
	pJob := carray[C.int32_t](p, P)
	pPS := carray[C.int32_t](p, P)
	pStatus := carray[C.int32_t](p, P)
	pNode := carray[C.int32_t](p, P)
	pFlags := carray[C.uint32_t](p, P)
	pTaskPrio := carray[C.int32_t](p, P)
	pCreated := carray[C.int64_t](p, P)
	pUID := carray[C.uint32_t](p, P)
	pClass := carray[C.int32_t](p, P)
	pNominated := carray[C.int32_t](p, P)
	pPortion := carray[C.double](p, P)
	pGpuMem := carray[C.int64_t](p, P)
	pGroup := carray[C.int32_t](p, P)
	copy(jUID, rankStrings(jids))
	sigIDs := map[string]int64{}
	podUIDs := make([]string, 0, P)
	classes := newStaticClasses(p.nodes) // pod classes by constraint sub-tree, node classes by the labels / taints those constraints see
	p.classes = classes
	pi, si := 0, 0
	for j, id := range jids {
		job := ci.PodGroupInfos[podgroup_info.PodGroupID(id)]
		p.jobs = append(p.jobs, job)
		jQueue[j] = -1
		if qi, ok := queueIdx[string(job.Queue)]; ok {
			jQueue[j] = C.int32_t(qi)
		}
		jPrio[j] = C.int32_t(job.Priority)
		if job.IsPreemptibleJob() {
			jPreempt[j] = 1
		}
		jCreated[j] = C.int64_t(job.CreationTimestamp.UnixNano())
		sig := string(job.GetSchedulingConstraintsSignature()) // only equality is used (actions/common/minimal_job_comparison.go:15-44)
		if _, ok := sigIDs[sig]; !ok {
			sigIDs[sig] = int64(len(sigIDs))
		}
		jSig[j] = C.int64_t(sigIDs[sig])
		if job.LastStartTimestamp != nil {
			jLastStart[j] = C.int64_t(job.LastStartTimestamp.UnixNano())
		}
		// pod-sets in name order; the rank is what PodSetOrderFn's name tie-break compares
		psNames := make([]string, 0, len(job.PodSets))
		for name := range job.PodSets {
			psNames = append(psNames, name)
		}
		sort.Strings(psNames)
		jFirstPS[j], jNPS[j] = C.int32_t(si), C.int32_t(len(psNames))
		psIndex := map[string]int{}
		for r, name := range psNames {
			psIndex[name] = si
			psJob[si], psMin[si], psRank[si] = C.int32_t(j), C.int32_t(job.PodSets[name].GetMinAvailable()), C.uint32_t(r)
			si++
		}
		tasks := make([]*pod_info.PodInfo, 0, len(job.GetAllPodsMap()))
		for _, t := range job.GetAllPodsMap() {
			tasks = append(tasks, t)
		}
		sort.Slice(tasks, func(a, b int) bool { return tasks[a].UID < tasks[b].UID })
		jFirstPod[j], jNPods[j] = C.int32_t(pi), C.int32_t(len(tasks))
		for _, t := range tasks {
			p.pods = append(p.pods, t)
			podUIDs = append(podUIDs, string(t.UID))
			req[0*P+pi] = C.double(t.ResReq.Cpu())
			req[1*P+pi] = C.double(t.ResReq.Memory())
			req[2*P+pi] = C.double(t.ResReq.GPUs()) // fractions: the portion rounded to 1/100 (gpu_resource_requirment.go:230-234)
			req[3*P+pi] = 1                          // pods := 1 (api/pod_info/pod_info.go:373-393)
			pJob[pi] = C.int32_t(j)
			pPS[pi] = C.int32_t(psIndex[t.SubGroupName])
			pStatus[pi] = statusBit(t.Status)
			pNode[pi], pNominated[pi], pGroup[pi] = -1, -1, -1
			if ni, ok := nodeIdx[t.NodeName]; ok {
				pNode[pi] = C.int32_t(ni)
			}
			if ni, ok := nodeIdx[t.Pod.Status.NominatedNodeName]; ok { // plugins/nominatednode/nominatednode.go:29-41
				pNominated[pi] = C.int32_t(ni)
			}
			if t.Pod.Spec.SchedulerName != params.schedulerName { // plugins/proportion/proportion.go:276-285
				pFlags[pi] |= C.KAI_POD_FOREIGN_SCHEDULER
			}
			if prio, ok := taskPriority(t); ok { // plugins/taskorder/task_order.go:28-63
				pFlags[pi] |= C.KAI_POD_HAS_TASK_PRIORITY
				pTaskPrio[pi] = C.int32_t(prio)
			}
			pFlags[pi] |= podModelFlags(t)
			if t.ResReq.IsFractionalRequest() && t.ResReq.GetNumOfGpuDevices() == 1 { // ABI v4 / v5: a fraction, or MiB, of ONE device
				if t.IsMemoryRequest() { // pod_info.go:463-468: GPUs() and the portion are 0, the request is the memory
					pGpuMem[pi] = C.int64_t(t.ResReq.GpuMemory())
				} else {
					pPortion[pi] = C.double(t.ResReq.GpuFractionalPortion())
				}
				if len(t.GPUGroups) > 0 {
					pGroup[pi] = C.int32_t(classes.groupID(t.NodeName, t.GPUGroups[0]))
				}
			}
			pCreated[pi] = C.int64_t(t.Pod.CreationTimestamp.UnixNano())
			pClass[pi] = C.int32_t(classes.podClass(t)) // NodeAffinity / nodeSelector / tolerations, canonicalised
			pi++
		}
	}
	copy(pUID, rankStrings(podUIDs))
	for i, n := range p.nodes {
		nodeClass[i] = C.int32_t(classes.nodeClass(n))
	}
	fit := classes.fitTable(p) // [pod classes][node classes], 1 = every static upstream Filter passes (k8s_internal/predicates/predicates.go:70-165)

	s.n_nodes, s.node_allocatable, s.node_flags, s.node_gpu_count, s.node_name_rank, s.node_class = C.int32_t(N), ptr(alloc), ptr(flags), ptr(gpuCount), ptr(cNodeRank), ptr(nodeClass)
	s.n_pods, s.pod_req, s.pod_job, s.pod_podset, s.pod_status, s.pod_node = C.int32_t(P), ptr(req), ptr(pJob), ptr(pPS), ptr(pStatus), ptr(pNode)
	s.pod_flags, s.pod_task_priority, s.pod_created_ns, s.pod_uid_rank, s.pod_class, s.pod_nominated_node = ptr(pFlags), ptr(pTaskPrio), ptr(pCreated), ptr(pUID), ptr(pClass), ptr(pNominated)
	s.n_podsets, s.podset_job, s.podset_min_available, s.podset_name_rank = C.int32_t(S), ptr(psJob), ptr(psMin), ptr(psRank)
	s.n_jobs, s.job_queue, s.job_priority, s.job_preemptible, s.job_created_ns, s.job_uid_rank = C.int32_t(J), ptr(jQueue), ptr(jPrio), ptr(jPreempt), ptr(jCreated), ptr(jUID)
	s.job_first_pod, s.job_n_pods, s.job_first_podset, s.job_n_podsets = ptr(jFirstPod), ptr(jNPods), ptr(jFirstPS), ptr(jNPS)
	s.n_queues, s.queue_parent, s.queue_priority, s.queue_created_ns, s.queue_uid_rank = C.int32_t(Q), ptr(qParent), ptr(qPrio), ptr(qCreated), ptr(qUID)
	s.queue_deserved, s.queue_limit, s.queue_oqw, s.queue_usage = ptr(qDeserved), ptr(qLimit), ptr(qOqw), ptr(qUsage)
	s.n_pod_classes, s.n_node_classes, s.class_fit = C.int32_t(classes.nPod()), C.int32_t(classes.nNode()), ptr(fit)
	s.job_signature, s.job_last_start_ns, s.queue_preempt_min_runtime_ns, s.queue_reclaim_min_runtime_ns = ptr(jSig), ptr(jLastStart), ptr(qPreMR), ptr(qRecMR)
	s.pod_gpu_portion, s.pod_gpu_group, s.node_gpu_memory, s.pod_gpu_memory = ptr(pPortion), ptr(pGroup), ptr(gpuMem), ptr(pGpuMem)
	packTopologies(p, ssn, nodeIdx) // Topology CRs -> node_domain / domain tables; RootSubGroupSet -> group tables (plugins/topology/topology_plugin.go:57-110)
	return p
}

// ------------------------------------------------------------------------------------------------ framework.Plugin
type packParams struct{ schedulerName, gpuWorkerLabel, cpuWorkerLabel string }

type plugin struct {
	params packParams
	pack   *packedSnapshot
}

var current *plugin

func New(args framework.PluginArguments) framework.Plugin {
	current = &plugin{params: packParams{schedulerName: args["schedulerName"], gpuWorkerLabel: args["gpuWorkerNodeLabelKey"], cpuWorkerLabel: args["cpuWorkerNodeLabelKey"]}}
	return current
}
func (p *plugin) Name() string { return "gpucore" }
func (p *plugin) OnSessionOpen(ssn *framework.Session) { // framework/interface.go:49-55
	p.pack = packSnapshot(ssn, p.params)
	if rc := C.kai_session_open(core, &p.pack.soa); rc != 0 {
		p.pack.fallback = true // e.g. KAI_ERR_UNSUPPORTED: leave this cycle to the Go actions
	}
}
func (p *plugin) OnSessionClose(*framework.Session) {
	C.kai_session_close(core)
	p.pack.free()
	p.pack = nil
}

// ------------------------------------------------------------------------------------------------ framework.Action
type action struct {
	kind     C.int
	name     framework.ActionType
	goAction framework.Action // the reference implementation: taken whenever the device path declines
}

func (a *action) Name() framework.ActionType { return a.name }
func (a *action) Execute(ssn *framework.Session) { // framework/interface.go:41-47
	if current == nil || current.pack == nil || current.pack.fallback {
		a.goAction.Execute(ssn)
		return
	}
	if zeroDepth[a.kind] {
		return // queueDepthPerAction 0: every leaf heap stays empty (kai_cgo_config.go), the action tries no job — neither here nor on the device
	}
	pack := current.pack
	capOps := C.int64_t(2*len(pack.pods) + 64)
	ops := (*C.kai_op)(C.malloc(C.size_t(capOps) * C.size_t(unsafe.Sizeof(C.kai_op{}))))
	defer C.free(unsafe.Pointer(ops))
	var n C.int64_t
	if rc := C.kai_action_execute(core, a.kind, ops, capOps, &n); rc != 0 {
		// The device may have applied part of the action before it failed (e.g. KAI_ERR_CAPACITY after some rounds), and in any case it will not see what
		// the Go action does now: the rest of this cycle stays on the Go actions.
		pack.fallback = true
		a.goAction.Execute(ssn)
		return
	}
	if !replay(ssn, a.name, pack, unsafe.Slice(ops, int(n))) {
		pack.fallback = true // the live session and the device state have parted: nothing more from the device in this cycle
	}
}

// replay: the committed operations through the real Statement.  kai_op.stmt numbers the Statements of the action in
// commit order; one id = one Statement, e.g. a reclaim "evict A, evict B, pipeline C" (framework/statement.go:536-575).
// An operation the live session refuses (the cache moved on since the snapshot, a bind conflict) discards its whole Statement —
// a gang is committed entirely or not at all — and ends the replay: false.
func replay(ssn *framework.Session, action framework.ActionType, pack *packedSnapshot, ops []C.kai_op) bool {
	groups := carray[C.int32_t](pack, len(pack.pods)) // PodInfo.GPUGroups[0] of the fraction pods after the action (freed with the snapshot)
	haveGroups := len(pack.pods) > 0 && C.kai_pod_gpu_groups(core, ptr(groups), C.int(len(pack.pods))) == 0
	for i := 0; i < len(ops); {
		stmt := ssn.Statement()
		id := ops[i].stmt
		var err error
		// EvictionMetadata of the Statement's evictions (actions/common/action.go:34-38): the gang = every task the solution evicts, the preemptor = the job the
		// Statement pipelines / allocates.  A Statement of a victim action holds one solution: "evict ..., pipeline the preemptor's tasks" (framework/statement.go:536-575).
		meta := eviction_info.EvictionMetadata{Action: string(action)}
		// The jobs re-placed behind the evictions are the victims' jobs AND the preemptor, in job order (actions/common/action.go:65-122), so the first pipelined task can be
		// a victim's: the preemptor is the job of a placed task that was Pending when the Statement began and that this Statement does not evict.
		// An elastic victim job may also place a pod of its own that was Pending in the same Statement (it is re-placed like any job of the scenario), and it can sort in front
		// of the preemptor: the preemptor is a job NONE of whose pods this Statement evicts.
		var preemptor *podgroup_info.PodGroupInfo
		evicted := map[C.int32_t]bool{}
		victimJob := map[common_info.PodGroupID]bool{}
		for k := i; k < len(ops) && ops[k].stmt == id; k++ {
			if ops[k].kind == C.KAI_OP_EVICT {
				meta.EvictionGangSize++
				evicted[ops[k].pod] = true
				victimJob[pack.pods[ops[k].pod].Job] = true
			}
		}
		for k := i; k < len(ops) && ops[k].stmt == id && preemptor == nil; k++ {
			task := pack.pods[ops[k].pod]
			if (ops[k].kind == C.KAI_OP_ALLOCATE || ops[k].kind == C.KAI_OP_PIPELINE) && !evicted[ops[k].pod] && !victimJob[task.Job] && task.Status == pod_status.Pending {
				preemptor = ssn.ClusterInfo.PodGroupInfos[task.Job]
			}
		}
		messages := map[int]string{} // getEvictionMessages: every message from the state BEFORE the first eviction (actions/common/action.go:51-60)
		if preemptor != nil {
			meta.Preemptor = &types.NamespacedName{Namespace: preemptor.Namespace, Name: preemptor.Name}
			for k := i; k < len(ops) && ops[k].stmt == id; k++ {
				if ops[k].kind == C.KAI_OP_EVICT {
					messages[k] = utils.GetMessageOfEviction(ssn, action, pack.pods[ops[k].pod], preemptor)
				}
			}
		}
		for ; i < len(ops) && ops[i].stmt == id; i++ {
			if err != nil {
				continue // skip the rest of a Statement that is going to be discarded
			}
			task := pack.pods[ops[i].pod]
			switch ops[i].kind {
			case C.KAI_OP_ALLOCATE, C.KAI_OP_PIPELINE:
				node := pack.nodes[ops[i].node]
				if haveGroups && task.IsSharedGPURequest() && groups[ops[i].pod] >= 0 {
					// SelectedGPUGroups of the BindRequest (cache/cache.go:290-330) come from PodInfo.GPUGroups: the group the device chose
					task.GPUGroups = []string{pack.classes.groupName(node.Name, int32(groups[ops[i].pod]))}
				}
				if ops[i].kind == C.KAI_OP_ALLOCATE {
					err = stmt.Allocate(task, node.Name)
				} else {
					err = stmt.Pipeline(task, node.Name, task.Status != pod_status.Pending)
				}
			case C.KAI_OP_EVICT:
				err = stmt.Evict(task, messages[i], meta)
			}
		}
		if err != nil {
			stmt.Discard()
			return false
		}
		if err = stmt.Commit(); err != nil {
			return false
		}
	}
	return true
}

func init() {
	framework.RegisterPluginBuilder("gpucore", New) // framework/plugins.go:31-47
	// framework/plugins.go:49-54: an action registered under the name of the original takes its place in the configured `actions:` list
	framework.RegisterAction(&action{kind: C.KAI_ACTION_ALLOCATE, name: framework.Allocate, goAction: allocate.New()})
	framework.RegisterAction(&action{kind: C.KAI_ACTION_CONSOLIDATION, name: framework.Consolidation, goAction: consolidation.New()})
	framework.RegisterAction(&action{kind: C.KAI_ACTION_RECLAIM, name: framework.Reclaim, goAction: reclaim.New()})
	framework.RegisterAction(&action{kind: C.KAI_ACTION_PREEMPT, name: framework.Preempt, goAction: preempt.New()})
}

// taskPriority, newStaticClasses (podClass / nodeClass / fitTable / groupID / groupName) and packTopologies: kai_cgo_classes.go.
