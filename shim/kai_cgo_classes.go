// kai_cgo_classes.go — the helpers of kai_cgo.go that look at live API objects: the task-priority label, the static upstream
// Filters compiled to a (pod class x node class) table, the shared-GPU group ids, and the Topology CRs / sub-group trees.
//
// They restate for *v1.Pod / *v1.Node / Topology objects what kai-scheduler_amd/csrc/kai_ingest.cpp does for snapshot.json
// (node_affinity_fits, taints_tolerated, the Topology walk); the C++ is what the tests pin (tests/test_ingest.py, among them the
// reference's ten AccumulatedNodeAffinities cases).  Here the matching itself is not restated at all: it calls the same helpers the
// upstream Filter plugins call (k8s.io/component-helpers), once per (pod class, node class) instead of once per (pod, node).
//
// Shipped as source (no Go toolchain in the build image of this repository), same package as kai_cgo.go.
package gpucore

/*
#include "kai_core.h"
*/
import "C"

import (
	"sort"
	"strconv"
	"strings"
	"unsafe"

	v1 "k8s.io/api/core/v1"
	"k8s.io/apimachinery/pkg/util/uuid"
	v1helper "k8s.io/component-helpers/scheduling/corev1"
	"k8s.io/component-helpers/scheduling/corev1/nodeaffinity"

	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/node_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/pod_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/podgroup_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/podgroup_info/subgroup_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/api/topology_info"
	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/framework"
)

// taskPriority: plugins/taskorder/task_order.go:28-63 reads the label kai.scheduler/task-priority; a value that does not parse
// counts as "no label" there.
func taskPriority(t *pod_info.PodInfo) (int, bool) {
	if t.Pod == nil {
		return 0, false
	}
	s, ok := t.Pod.Labels["kai.scheduler/task-priority"]
	if !ok {
		return 0, false
	}
	v, err := strconv.Atoi(s)
	if err != nil {
		return 0, false
	}
	return v, true
}

// ------------------------------------------------------------------------------------------------ static predicate classes (SURVEY §8f n4)
// Pods fall into classes by the part of their spec the static Filters read (nodeSelector, required node affinity, tolerations);
// nodes by what those constraints can see: the values of the label keys any pod constraint names, the NoSchedule / NoExecute
// taints, and the node name when some term uses matchFields.  class_fit[pod class][node class] = NodeAffinity and TaintToleration
// both pass (k8s_internal/predicates/predicates.go:70-165 wires exactly these as per-node Filters without cluster state).
type staticClasses struct {
	nodes      []*node_info.NodeInfo
	podIDs     map[string]int
	podReps    []*v1.Pod
	usedKeys   map[string]struct{}
	usesName   bool
	nodeIDs    map[string]int
	nodeReps   []*v1.Node
	nodeOf     map[string]int // node name -> class, filled by nodeClass
	groupIDs   map[string]int32   // node + "\x00" + group name -> id
	groupNames map[groupKey]string // (node, id) -> group name: the snapshot's names and the UUIDs drawn in replay for devices opened in this cycle
	nextNewGrp int32
}

type groupKey struct {
	node string
	id   int32
}

func newStaticClasses(nodes []*node_info.NodeInfo) *staticClasses {
	return &staticClasses{nodes: nodes, podIDs: map[string]int{}, usedKeys: map[string]struct{}{}, nodeIDs: map[string]int{}, nodeOf: map[string]int{},
		groupIDs: map[string]int32{}, groupNames: map[groupKey]string{}, nextNewGrp: 1 << 20} // KAI_NEW_GROUP: ids of non-numeric group names (kai_engine.hpp)
}

func reqSig(rs []v1.NodeSelectorRequirement) string {
	parts := make([]string, 0, len(rs))
	for _, r := range rs {
		vals := append([]string(nil), r.Values...)
		sort.Strings(vals)
		parts = append(parts, r.Key+"\x01"+string(r.Operator)+"\x01"+strings.Join(vals, "\x02"))
	}
	sort.Strings(parts) // requirements of one term are ANDed: their order does not matter
	return strings.Join(parts, "\x03")
}

// podClass: canonical signature of the constraint sub-trees; two pods with the same signature pass the same nodes
func (c *staticClasses) podClass(t *pod_info.PodInfo) int {
	spec := &t.Pod.Spec
	var b strings.Builder
	keys := make([]string, 0, len(spec.NodeSelector))
	for k := range spec.NodeSelector {
		keys = append(keys, k)
	}
	sort.Strings(keys)
	for _, k := range keys {
		b.WriteString(k + "=" + spec.NodeSelector[k] + "\x04")
		c.usedKeys[k] = struct{}{}
	}
	b.WriteString("\x05")
	if spec.Affinity != nil && spec.Affinity.NodeAffinity != nil && spec.Affinity.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil {
		terms := spec.Affinity.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms
		ts := make([]string, 0, len(terms))
		for _, term := range terms {
			for _, r := range term.MatchExpressions {
				c.usedKeys[r.Key] = struct{}{}
			}
			if len(term.MatchFields) > 0 {
				c.usesName = true
			}
			ts = append(ts, reqSig(term.MatchExpressions)+"\x06"+reqSig(term.MatchFields))
		}
		sort.Strings(ts) // terms are ORed
		b.WriteString("R" + strings.Join(ts, "\x07"))
	}
	b.WriteString("\x05")
	tols := make([]string, 0, len(spec.Tolerations))
	for _, tol := range spec.Tolerations {
		tols = append(tols, tol.Key+"\x01"+string(tol.Operator)+"\x01"+tol.Value+"\x01"+string(tol.Effect))
	}
	sort.Strings(tols)
	b.WriteString(strings.Join(tols, "\x08"))
	sig := b.String()
	if id, ok := c.podIDs[sig]; ok {
		return id
	}
	id := len(c.podReps)
	c.podIDs[sig] = id
	c.podReps = append(c.podReps, t.Pod)
	return id
}

// nodeClass must be called after every pod went through podClass (the label keys in use are known only then)
func (c *staticClasses) nodeClass(n *node_info.NodeInfo) int {
	keys := make([]string, 0, len(c.usedKeys))
	for k := range c.usedKeys {
		keys = append(keys, k)
	}
	sort.Strings(keys)
	var b strings.Builder
	for _, k := range keys {
		if v, ok := n.Node.Labels[k]; ok {
			b.WriteString(k + "=" + v + "\x02")
		} else {
			b.WriteString(k + "\x01\x02")
		}
	}
	for _, t := range n.Node.Spec.Taints {
		if t.Effect == v1.TaintEffectNoSchedule || t.Effect == v1.TaintEffectNoExecute { // the effects the TaintToleration Filter looks at
			b.WriteString(t.Key + "\x03" + t.Value + "\x03" + string(t.Effect) + "\x04")
		}
	}
	if c.usesName {
		b.WriteString("\x05" + n.Name)
	}
	sig := b.String()
	id, ok := c.nodeIDs[sig]
	if !ok {
		id = len(c.nodeReps)
		c.nodeIDs[sig] = id
		c.nodeReps = append(c.nodeReps, n.Node)
	}
	c.nodeOf[n.Name] = id
	return id
}

func (c *staticClasses) nPod() int  { return max(1, len(c.podReps)) }
func (c *staticClasses) nNode() int { return max(1, len(c.nodeReps)) }

// fitTable: [pod classes][node classes] in C memory, 1 = every static upstream Filter passes
func (c *staticClasses) fitTable(p *packedSnapshot) []C.uint8_t {
	pc, nc := c.nPod(), c.nNode()
	fit := carray[C.uint8_t](p, pc*nc)
	for i := range fit {
		fit[i] = 1
	}
	for a, pod := range c.podReps {
		required := nodeaffinity.GetRequiredNodeAffinity(pod) // what the NodeAffinity plugin's Filter evaluates (nodeSelector AND required terms)
		for b, node := range c.nodeReps {
			ok, _ := required.Match(node)
			if ok {
				_, untolerated := v1helper.FindMatchingUntoleratedTaint(node.Spec.Taints, pod.Spec.Tolerations, func(t *v1.Taint) bool {
					return t.Effect == v1.TaintEffectNoSchedule || t.Effect == v1.TaintEffectNoExecute // tainttoleration.Filter
				})
				ok = !untolerated
			}
			if !ok {
				fit[a*nc+b] = 0
			}
		}
	}
	return fit
}

// groupID: PodInfo.GPUGroups[0] as the int32 the ABI carries.  Numeric names (the device index labels of the reservation pods) keep
// their value; any other name (a UUID) gets an id from KAI_NEW_GROUP on, per (node, name) — plugins/predicates/predicates.go:320-330
// only asks whether a name is "new", and groups never span nodes.
func (c *staticClasses) groupID(nodeName, group string) int32 {
	if v, err := strconv.Atoi(group); err == nil && v >= 0 && v < 1<<20 {
		return int32(v)
	}
	key := nodeName + "\x00" + group
	if id, ok := c.groupIDs[key]; ok {
		return id
	}
	id := c.nextNewGrp
	c.nextNewGrp++
	c.groupIDs[key] = id
	c.groupNames[groupKey{nodeName, id}] = group
	return id
}

// groupName: the inverse, for the GPU groups kai_pod_gpu_groups reports after an action (SelectedGPUGroups of a BindRequest).
// An id >= KAI_NEW_GROUP that no snapshot name maps to is a device the engine opened in THIS cycle.  The reference names such a
// group in the scheduler, before stmt.Allocate / Pipeline: findGpuForSharingOnNode draws uuid.NewUUID() (gpu_sharing/gpuSharing.go:73-83),
// cache.go:316 copies PodInfo.GPUGroups into BindRequest.SelectedGPUGroups.  So the first pod that lands in (node, id) draws the
// UUID and every later pod of the same (node, id) — the device put it in the same group — reads it back from groupIDs.
func (c *staticClasses) groupName(nodeName string, id int32) string {
	if id < 1<<20 {
		return strconv.Itoa(int(id))
	}
	if name, ok := c.groupNames[groupKey{nodeName, id}]; ok {
		return name
	}
	name := string(uuid.NewUUID())
	c.groupIDs[nodeName+"\x00"+name] = id
	c.groupNames[groupKey{nodeName, id}] = name
	return name
}

// ------------------------------------------------------------------------------------------------ topologies + sub-group trees
// Topology CRs -> level rows, node_domain, the domain table; every job's RootSubGroupSet -> the group tables
// (plugins/topology/topology_plugin.go:57-110, topology_structs.go:94-101, api/podgroup_info/subgroup_info/*.go).
func packTopologies(p *packedSnapshot, ssn *framework.Session, nodeIdx map[string]int) {
	s := &p.soa
	N := len(p.nodes)
	topos := ssn.ClusterInfo.Topologies
	sort.Slice(topos, func(a, b int) bool { return topos[a].Name < topos[b].Name })
	T := len(topos)
	levelOff := carray[C.int32_t](p, T+1)
	topoIdx := map[string]int{}
	for t, tp := range topos {
		topoIdx[tp.Name] = t
		levelOff[t+1] = levelOff[t] + C.int32_t(len(tp.Spec.Levels))
	}
	TL := int(levelOff[T])
	nodeDomain := carray[C.int32_t](p, max(TL, 1)*max(N, 1))
	for i := range nodeDomain {
		nodeDomain[i] = -1
	}
	type dom struct {
		topo, level, parent int
		id                  string
	}
	var doms []dom
	for t, tp := range topos {
		ids := map[string]int{}
		for ni, n := range p.nodes { // engine order is re-derived by the library from node_name_rank; here: ABI node order
			vals := make([]string, 0, len(tp.Spec.Levels))
			complete := true
			for _, lv := range tp.Spec.Levels { // a node joins a topology only with every level label (topology/common.go:70-77)
				v, ok := n.Node.Labels[lv.NodeLabel]
				if !ok {
					complete = false
					break
				}
				vals = append(vals, v)
			}
			if !complete {
				continue
			}
			parent := -1
			for l := range vals {
				id := strings.Join(vals[:l+1], ".") // DomainID (topology_structs.go:94-101)
				d, ok := ids[strconv.Itoa(l)+"\x00"+id]
				if !ok {
					d = len(doms)
					ids[strconv.Itoa(l)+"\x00"+id] = d
					doms = append(doms, dom{t, int(levelOff[t]) + l, parent, id})
				}
				nodeDomain[(int(levelOff[t])+l)*N+ni] = C.int32_t(d)
				parent = d
			}
		}
	}
	D := len(doms)
	domLevel := carray[C.int32_t](p, D)
	domParent := carray[C.int32_t](p, D)
	domRank := carray[C.uint32_t](p, D)
	for t := range topos { // ranks of the ID strings inside a topology (sortTree's tie-break, job_filtering.go:470-477)
		var idx []int
		var ss []string
		for d, x := range doms {
			if x.topo == t {
				idx = append(idx, d)
				ss = append(ss, x.id)
			}
		}
		for k, r := range rankStrings(ss) {
			domRank[idx[k]] = r
		}
	}
	for d, x := range doms {
		domLevel[d], domParent[d] = C.int32_t(x.level), C.int32_t(x.parent)
	}
	s.n_topologies, s.topo_level_off, s.n_topo_levels, s.node_domain = C.int32_t(T), ptr(levelOff), C.int32_t(TL), ptr(nodeDomain)
	s.n_domains, s.domain_level, s.domain_parent, s.domain_id_rank = C.int32_t(D), ptr(domLevel), ptr(domParent), ptr(domRank)

	// constraint -> (topology | -1 none | -2 missing, required level, preferred level); an unknown level name lies beyond the last level
	constraint := func(tc *topology_info.TopologyConstraintInfo) (C.int32_t, C.int32_t, C.int32_t) {
		if tc == nil || tc.Topology == "" {
			return -1, -1, -1
		}
		t, ok := topoIdx[tc.Topology]
		if !ok {
			return -2, -1, -1
		}
		level := func(name string) C.int32_t {
			if name == "" {
				return -1
			}
			for l, lv := range topos[t].Spec.Levels {
				if lv.NodeLabel == name {
					return C.int32_t(l)
				}
			}
			return 1000000
		}
		return C.int32_t(t), level(tc.RequiredLevel), level(tc.PreferredLevel)
	}

	// groups: pre-order walk of every job's RootSubGroupSet; pod-sets were laid out in name order by packSnapshot
	var gJob, gParent, gTopo, gReq, gPref []C.int32_t
	var gNames []string
	var gNameJob []int
	S := int(s.n_podsets)
	psGroup := carray[C.int32_t](p, S)
	psTopo := carray[C.int32_t](p, S)
	psReq := carray[C.int32_t](p, S)
	psPref := carray[C.int32_t](p, S)
	jRoot := carray[C.int32_t](p, len(p.jobs))
	firstPS := unsafeSlice(s.job_first_podset, len(p.jobs))
	for j, job := range p.jobs {
		psNames := make([]string, 0, len(job.PodSets))
		for name := range job.PodSets {
			psNames = append(psNames, name)
		}
		sort.Strings(psNames)
		psIndex := map[string]int{}
		for r, name := range psNames {
			psIndex[name] = int(firstPS[j]) + r
		}
		var walk func(g *subgroup_info.SubGroupSet, parent int) int
		walk = func(g *subgroup_info.SubGroupSet, parent int) int {
			me := len(gJob)
			t, rq, pf := constraint(g.GetTopologyConstraint())
			gJob, gParent, gTopo, gReq, gPref = append(gJob, C.int32_t(j)), append(gParent, C.int32_t(parent)), append(gTopo, t), append(gReq, rq), append(gPref, pf)
			gNames, gNameJob = append(gNames, g.GetName()), append(gNameJob, j)
			for _, child := range g.GetChildGroups() {
				walk(child, me)
			}
			for _, ps := range g.GetChildPodSets() {
				si := psIndex[ps.GetName()]
				psGroup[si] = C.int32_t(me)
				psTopo[si], psReq[si], psPref[si] = constraint(ps.GetTopologyConstraint())
			}
			return me
		}
		jRoot[j] = C.int32_t(walk(rootSubGroupSet(job), -1))
	}
	G := len(gJob)
	cJob, cParent, cTopo, cReq, cPref := carray[C.int32_t](p, G), carray[C.int32_t](p, G), carray[C.int32_t](p, G), carray[C.int32_t](p, G), carray[C.int32_t](p, G)
	copy(cJob, gJob)
	copy(cParent, gParent)
	copy(cTopo, gTopo)
	copy(cReq, gReq)
	copy(cPref, gPref)
	cRank := carray[C.uint32_t](p, G)
	for j := range p.jobs { // rank of the SubGroupSet name inside its job (framework/session_plugins.go:273-282)
		var idx []int
		var ss []string
		for g := range gNames {
			if gNameJob[g] == j {
				idx = append(idx, g)
				ss = append(ss, gNames[g])
			}
		}
		for k, r := range rankStrings(ss) {
			cRank[idx[k]] = r
		}
	}
	s.n_groups, s.group_job, s.group_parent, s.group_name_rank = C.int32_t(G), ptr(cJob), ptr(cParent), ptr(cRank)
	s.group_topology, s.group_required_level, s.group_preferred_level, s.job_root_group = ptr(cTopo), ptr(cReq), ptr(cPref), ptr(jRoot)
	s.podset_group, s.podset_topology, s.podset_required_level, s.podset_preferred_level = ptr(psGroup), ptr(psTopo), ptr(psReq), ptr(psPref)
}

func rootSubGroupSet(job *podgroup_info.PodGroupInfo) *subgroup_info.SubGroupSet { return job.RootSubGroupSet }

func unsafeSlice(p *C.int32_t, n int) []C.int32_t {
	if p == nil || n == 0 {
		return nil
	}
	return unsafe.Slice(p, n)
}
