// kai_cgo_config.go — conf.SchedulerConfiguration + conf.SchedulerParams -> kai_config, what Init takes.  The same mapping as kai_ingest.cpp applies to a snapshot
// file's "config" / "schedulerParams" (tests/test_ingest.py pins that one on the reference's conf_util tests); kept here as source for the reference-side build.
// Shipped as source (no Go toolchain in the build image of this repository), same package as kai_cgo.go.
package gpucore

/*
#include "kai_core.h"
*/
import "C"

import (
	"strconv"
	"strings"
	"time"

	"github.com/NVIDIA/KAI-scheduler/pkg/scheduler/conf"
)

// plugins of the default tier list that are on the device path (conf_util/scheduler_conf_util.go:36-61); the others (kubeflow, ray, snapshot, podaffinity,
// dynamicresources) either do not score / order on this path or send their pods to the fallback flag.
var pluginBits = map[string]C.uint32_t{
	"predicates": C.KAI_PLUGIN_PREDICATES, "proportion": C.KAI_PLUGIN_PROPORTION, "priority": C.KAI_PLUGIN_PRIORITY, "elastic": C.KAI_PLUGIN_ELASTIC,
	"nodeavailability": C.KAI_PLUGIN_NODEAVAILABILITY, "resourcetype": C.KAI_PLUGIN_RESOURCETYPE, "subgrouporder": C.KAI_PLUGIN_SUBGROUPORDER,
	"taskorder": C.KAI_PLUGIN_TASKORDER, "nominatednode": C.KAI_PLUGIN_NOMINATEDNODE, "nodeplacement": C.KAI_PLUGIN_NODEPLACEMENT,
	"minruntime": C.KAI_PLUGIN_MINRUNTIME, "topology": C.KAI_PLUGIN_TOPOLOGY, "gpusharingorder": C.KAI_PLUGIN_GPUSHARINGORDER,
	"gpupack": C.KAI_PLUGIN_GPUPACK, "gpuspread": C.KAI_PLUGIN_GPUSPREAD,
}

// actions whose queueDepthPerAction entry is an explicit 0 (filled by ConfigFromScheduler): they pop no job in the reference
var zeroDepth = map[C.int]bool{}

var actionIndex = map[string]int{"allocate": C.KAI_ACTION_ALLOCATE, "consolidation": C.KAI_ACTION_CONSOLIDATION, "reclaim": C.KAI_ACTION_RECLAIM, "preempt": C.KAI_ACTION_PREEMPT}

// ConfigFromScheduler fills kai_config from the scheduler's configuration.  Returns the actions of conf.Actions that the device path implements, in order
// (stalegangeviction is not a placement action and stays on the Go side).
func ConfigFromScheduler(sc *conf.SchedulerConfiguration, params conf.SchedulerParams, now time.Time) (C.kai_config, []string) {
	var cfg C.kai_config
	cfg.abi_version = C.KAI_ABI_VERSION
	cfg.gpu_strategy, cfg.cpu_strategy = C.KAI_BINPACK, C.KAI_BINPACK // nodeplacement.go:59-70 defaults
	cfg.k_value, cfg.reclaimer_saturation_multiplier = 1.0, 1.0         // proportion.go:67-93 defaults
	cfg.min_node_gpu_memory = 100                                       // cluster_info.go:242-257 as written (SURVEY Appendix D)
	b2i := func(b bool) C.int32_t {
		if b {
			return 1
		}
		return 0
	}
	cfg.restrict_node_scheduling = b2i(params.RestrictSchedulingNodes)
	cfg.max_consolidation_preemptees = C.int32_t(params.MaxNumberConsolidationPreemptees)
	cfg.use_scheduling_signatures = b2i(params.UseSchedulingSignatures)
	cfg.allow_consolidating_reclaim = b2i(params.AllowConsolidatingReclaim)
	cfg.full_hierarchy_fairness = b2i(params.FullHierarchyFairness)
	cfg.now_ns = C.int64_t(now.UnixNano())
	for i := range cfg.queue_depth { // framework/session.go:398-404: no entry = every job of the queue
		cfg.queue_depth[i] = -1
	}
	zeroDepth = map[C.int]bool{}
	for name, depth := range sc.QueueDepthPerAction {
		if i, ok := actionIndex[name]; ok {
			cfg.queue_depth[i] = C.int32_t(depth)
			// An explicit 0 is NOT "infinite": PriorityQueue.Push removes index 0 — the element just pushed or the best one — whenever Len() > 0
			// (scheduler_util/priority_queue.go:50-55), so every leaf heap of the action stays empty and the action pops no job.  The ABI reads
			// 0 as "no limit", so the shim keeps such an action away from the device and runs it as the no-op it is (action.Execute).
			if depth == 0 {
				zeroDepth[C.int(i)] = true
			}
		}
	}
	if len(sc.Tiers) == 0 {
		cfg.plugins = C.KAI_PLUGIN_ALL
	}
	for _, tier := range sc.Tiers {
		for _, pl := range tier.Plugins {
			cfg.plugins |= pluginBits[pl.Name]
			switch pl.Name {
			case "nodeplacement":
				if pl.Arguments["gpu"] == "spread" {
					cfg.gpu_strategy = C.KAI_SPREAD
				}
				if pl.Arguments["cpu"] == "spread" {
					cfg.cpu_strategy = C.KAI_SPREAD
				}
			case "proportion":
				if v, err := strconv.ParseFloat(pl.Arguments["kValue"], 64); err == nil {
					if v <= 0 { // proportion.go:81-84: "kValue must be > 0.0 ... Setting as 0"
						v = 0
					}
					cfg.k_value = C.double(v)
				}
				if v, err := strconv.ParseFloat(pl.Arguments["relcaimerSaturationMultiplier"], 64); err == nil && v >= 1.0 { // (the argument's spelling in proportion.go)
					cfg.reclaimer_saturation_multiplier = C.double(v)
				}
			case "minruntime": // minruntime.go:40-70
				if d, err := time.ParseDuration(pl.Arguments["defaultPreemptMinRuntime"]); err == nil && d >= 0 {
					cfg.default_preempt_min_runtime_ns = C.int64_t(d.Nanoseconds())
				}
				if d, err := time.ParseDuration(pl.Arguments["defaultReclaimMinRuntime"]); err == nil && d >= 0 {
					cfg.default_reclaim_min_runtime_ns = C.int64_t(d.Nanoseconds())
				}
				if pl.Arguments["reclaimResolveMethod"] == "queue" {
					cfg.reclaim_resolve_method = 1
				}
			}
		}
	}
	var actions []string
	for _, a := range strings.Split(sc.Actions, ",") {
		a = strings.TrimSpace(a)
		if _, ok := actionIndex[a]; ok {
			actions = append(actions, a)
		}
	}
	return cfg, actions
}
