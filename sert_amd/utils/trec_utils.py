"""TREC topic / run I/O (cvangysel.trec_utils call sites: bin/query.py:72, :121,
:151, :156) and a trec_eval-compatible NDCG@k / MAP evaluator (trec_eval is not
available offline; product-search.sh:149-170 uses it to pick the best epoch).

Topic files are ``id;space separated terms`` per line
(resources/product-search/*/topics).  Run files are the 6-column TREC format
``topic Q0 entity rank score tag``.  qrels are ``topic 0 entity relevance``.
"""
import collections
import math
import re


def parse_topics(file_or_files, delimiter=';'):
    """-> OrderedDict topic_id -> raw topic text."""
    if not isinstance(file_or_files, (list, tuple)):
        file_or_files = [file_or_files]
    topics = collections.OrderedDict()
    for f in file_or_files:
        for line in f:
            line = line.rstrip('\n')
            if not line.strip():
                continue
            topic_id, _, text = line.partition(delimiter)
            topic_id = topic_id.strip()
            if topic_id in topics:
                raise RuntimeError('Duplicate topic "{0}".'.format(topic_id))
            topics[topic_id] = text.strip()
    return topics


_TOKEN = re.compile(r"[^\W_]+(?:['\-][^\W_]+)*", re.UNICODE)


def parse_query(text):
    """Lower-cased word tokens; numeric tokens are kept (query.py:120)."""
    return [t.lower() for t in _TOKEN.findall(text)]


def write_run(model_name, data, out_f, max_objects_per_query=None):
    """data: query_id -> iterable of (score, object_id).  Ranked by score descending; ties by
    object id DESCENDING -- the order trec_eval itself imposes when it re-sorts a run (it ignores
    the rank column), so the rank written here is the rank that gets evaluated (see _ranked)."""
    for query_id in data:
        ranked = sorted(data[query_id], key=lambda so: (float(so[0]), str(so[1])), reverse=True)
        if max_objects_per_query:
            ranked = ranked[:max_objects_per_query]
        for rank, (score, object_id) in enumerate(ranked, 1):
            out_f.write(u'{0} Q0 {1} {2} {3:.10f} {4}\n'.format(
                query_id, object_id, rank, float(score), model_name))


def parse_qrels(f):
    """-> dict topic -> dict entity -> relevance (float)."""
    qrels = collections.defaultdict(dict)
    for line in f:
        parts = line.split()
        if len(parts) < 4:
            continue
        qrels[parts[0]][parts[2]] = float(parts[3])
    return qrels


def parse_run(f):
    """-> dict topic -> list of (score, entity), file order."""
    run = collections.defaultdict(list)
    for line in f:
        parts = line.split()
        if len(parts) < 6:
            continue
        run[parts[0]].append((float(parts[4]), parts[2]))
    return run


def _ranked(entries):
    # trec_eval ignores the rank column: score desc, then doc id DESC
    # (trec_eval sorts ties by docno in reverse lexicographic order)
    return [e for _, e in sorted(entries, key=lambda se: (se[0], se[1]), reverse=True)]


def ndcg_at_k(ranked_entities, relevance, k=100):
    gains = [relevance.get(e, 0.0) for e in ranked_entities[:k]]
    dcg = sum(g / math.log2(i + 2) for i, g in enumerate(gains))
    ideal = sorted((r for r in relevance.values() if r > 0), reverse=True)[:k]
    idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal))
    return dcg / idcg if idcg > 0 else 0.0


def average_precision(ranked_entities, relevance):
    num_rel = sum(1 for r in relevance.values() if r > 0)
    if num_rel == 0:
        return 0.0
    hits, total = 0, 0.0
    for i, e in enumerate(ranked_entities, 1):
        if relevance.get(e, 0.0) > 0:
            hits += 1
            total += hits / float(i)
    return total / num_rel


def evaluate_run(run, qrels, k=100):
    """Mean NDCG@k and MAP over the topics that have relevance judgements."""
    ndcgs, aps = [], []
    for topic, rel in qrels.items():
        ranked = _ranked(run.get(topic, []))
        ndcgs.append(ndcg_at_k(ranked, rel, k))
        aps.append(average_precision(ranked, rel))
    n = max(1, len(ndcgs))
    return {'ndcg_cut_%d' % k: sum(ndcgs) / n, 'map': sum(aps) / n, 'num_q': len(ndcgs)}
