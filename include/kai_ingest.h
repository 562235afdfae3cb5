/*
 * kai_ingest.h — C ABI of libkai_ingest: reference-schema snapshot (snapshot.json / snapshot.zip) → kai_snapshot_soa.
 *
 * This is SURVEY.md §8f row n1 (snapshot ingest) and n4 (static upstream predicates compiled to class tables): the step
 * immediately BEFORE the placement path.  It replaces, for replayed snapshots, what the reference does in
 *   cmd/snapshot-tool/main.go:118-257          (load snapshot.zip, feed fake clientsets)
 *   pkg/scheduler/cache/cluster_info/cluster_info.go:118-228   (ClusterInfo.Snapshot: nodes, pods, queues, pod groups)
 *   pkg/scheduler/api/pod_info/pod_info.go:172-214,373-445     (NewTaskInfo, getPodResourceRequest, getTaskStatus)
 *   pkg/scheduler/api/node_info/node_info.go:105-156           (NewNodeInfo)
 *   pkg/scheduler/api/queue_info/queue_info.go:45-90, cache/cluster_info/queue.go:53-129 (queues, hierarchy, orphans)
 *   pkg/scheduler/api/podgroup_info/job_info.go:160-251, subgroup_info/factory.go:16-135  (pod groups, sub-group tree)
 *   pkg/scheduler/conf_util/scheduler_conf_util.go:36-107, conf/scheduler_conf.go:31-88   (actions, tiers, params)
 * and pre-evaluates the upstream kube-scheduler Filters that depend only on (pod spec, node object) — NodeAffinity
 * (nodeSelector + required node affinity) and TaintToleration, k8s.io/kubernetes v1.34.2
 * pkg/scheduler/framework/plugins/{nodeaffinity,tainttoleration}, called from
 * pkg/scheduler/k8s_internal/predicates/predicates.go:70-165 — into the pod_class x node_class table of the ABI.
 *
 * Pure host code (no HIP, no device): a data-format conversion.  The output feeds kai_session_open of kai_core.h
 * unchanged; placement itself never runs here.
 *
 * Conventions: 0 = ok, negative = kai_status (kai_core.h).  The handle owns every buffer the returned structs point to.
 */
#ifndef KAI_INGEST_H
#define KAI_INGEST_H

#include <stddef.h>
#include <stdint.h>

#include "kai_core.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kai_ingest kai_ingest; /* opaque */

typedef struct kai_ingest_options {
    const char* scheduler_name; /* NULL: schedulerParams.schedulerName, else "kai-scheduler" (pkg/common/constants/constants.go:18) */
    int64_t now_ns;             /* "now" of the cycle for the minruntime plugin; 0: the latest timestamp found in the snapshot */
    int32_t reserved[4];
} kai_ingest_options;

/* name tables kept on the host (strings never cross kai_core's ABI) */
typedef enum kai_ingest_kind { KAI_NAME_NODE = 0, KAI_NAME_POD = 1, KAI_NAME_JOB = 2, KAI_NAME_QUEUE = 3, KAI_NAME_PODSET = 4, KAI_NAME_RESOURCE = 5 } kai_ingest_kind;

/* parse a snapshot.json held in memory */
int kai_ingest_parse(const char* json, size_t len, const kai_ingest_options* opt, kai_ingest** out);
/* load snapshot.json, or a snapshot.zip holding it (plugins/snapshot/snapshot.go:33, cmd/snapshot-tool/main.go:118-147) */
int kai_ingest_load(const char* path, const kai_ingest_options* opt, kai_ingest** out);

const kai_snapshot_soa* kai_ingest_snapshot(const kai_ingest* h);
const kai_config* kai_ingest_config(const kai_ingest* h);
/* the configured action list in order, as kai_action values; "stalegangeviction" (not a placement action) is skipped.
 * returns the count (may exceed cap), or KAI_ERR_INVALID_ARG for an action name the reference does not register */
int kai_ingest_actions(const kai_ingest* h, int32_t* out, int cap);
/* name of object idx of the given kind (pods: "namespace/name"), NULL when out of range */
const char* kai_ingest_name(const kai_ingest* h, int kind, int idx);
/* newline-separated notes: objects dropped, features routed to the CPU fallback, ignored plugins (never NULL) */
const char* kai_ingest_warnings(const kai_ingest* h);
/* The committed operations of kai_action_execute written as what the reference's cache creates for them — SURVEY §8f n3, the data format
 * AFTER the path: {"bindRequests": [scheduling.run.ai/v1alpha2 BindRequest, one per Allocate, in commit order — cache/cache.go:290-330
 * createBindRequest: name / namespace / owner reference of the pod, label selected-node (+ the node-pool label), spec.podName, selectedNode,
 * receivedResourceType "Regular", receivedGPU{count, portion "%.2f"} per node_info.go:746-768], "evictions": [pods handed to cache.Evict,
 * cache.go:216-252, with their pod group], "pipelined": [pod → node; no cluster side effect, framework/statement.go:197-295]}.
 * One call hands over the whole batch: 10^6 placements are one document, not 10^6 API creates.  Writes a NUL-terminated string; *len = its
 * length.  KAI_ERR_CAPACITY (with *len set) when cap < *len + 1; out may be NULL to size the buffer. */
int kai_ingest_decisions_json(const kai_ingest* h, const kai_op* ops, int64_t n_ops, char* out, size_t cap, size_t* len);
void kai_ingest_free(kai_ingest* h);
/* detail of the last failed parse/load on this thread (never NULL) */
const char* kai_ingest_last_error(void);

/* k8s.io/apimachinery resource.Quantity known-answer hooks (api/resource_info/resource_vector.go:185-190 uses exactly these):
 * MilliValue() and Value() round UP.  Return 0 or KAI_ERR_INVALID_ARG. */
int kai_quantity_milli(const char* s, int64_t* out);
int kai_quantity_value(const char* s, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* KAI_INGEST_H */
