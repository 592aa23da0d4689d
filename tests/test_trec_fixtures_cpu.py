"""CPU: topic / qrel parsing and the in-repo nDCG@100 / MAP evaluator on DATA cut from the
reference's own benchmark resources (resources/product-search/home_and_kitchen; fixture script:
tests/golden/make_resource_fixtures.py).  The evaluator stands in for trec_eval
(product-search.sh:149-170), which is not available offline."""
import io
import math
import os

import numpy as np

from sert_amd.utils import trec_utils as T

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'product_search')


def _load():
    with open(os.path.join(HERE, 'topics')) as f:
        topics = T.parse_topics(f)
    with open(os.path.join(HERE, 'qrel_test')) as f:
        qrels = T.parse_qrels(f)
    with open(os.path.join(HERE, 'product_list')) as f:
        products = [line.strip() for line in f if line.strip()]
    return topics, qrels, products


def test_reference_topics_and_qrels_parse():
    topics, qrels, products = _load()
    assert len(topics) == 60 and list(topics)[:3] == ['0', '1', '2']
    assert topics['0'] == 'bathroom mats squeegees bath shower stall'        # resources/.../topics line 1
    lengths = [len(T.parse_query(text)) for text in topics.values()]
    assert min(lengths) >= 1 and max(lengths) <= 20
    assert all(t in topics for t in qrels)
    assert all(rel == 1.0 for judged in qrels.values() for rel in judged.values())   # binary judgements
    assert set(e for judged in qrels.values() for e in judged) <= set(products)


def test_evaluator_on_reference_judgements():
    topics, qrels, products = _load()
    rng = np.random.RandomState(0)
    # (1) the ideal run: every relevant entity first -> nDCG = MAP = 1
    ideal = {t: [(1.0 - i * 1e-3, e) for i, e in enumerate(sorted(judged))] for t, judged in qrels.items()}
    res = T.evaluate_run(ideal, qrels, k=100)
    assert abs(res['ndcg_cut_100'] - 1.0) < 1e-12 and abs(res['map'] - 1.0) < 1e-12 and res['num_q'] == len(qrels)
    # (2) a run through write_run / parse_run: relevant entities at known ranks, hand-computed metrics
    data, expect_ndcg, expect_ap = {}, [], []
    for t, judged in qrels.items():
        rel = sorted(judged)
        others = [p for p in products if p not in judged][:150]
        rng.shuffle(others)
        ranking = others[:]
        ranks = sorted(rng.choice(np.arange(1, 121), size=len(rel), replace=False))   # 1-based ranks of the relevant ones
        for r, e in zip(ranks, rel):
            ranking.insert(r - 1, e)
        data[t] = [(1000.0 - i, e) for i, e in enumerate(ranking)]
        final_ranks = [ranking.index(e) + 1 for e in rel]
        dcg = sum(1.0 / math.log2(r + 1) for r in final_ranks if r <= 100)
        idcg = sum(1.0 / math.log2(i + 2) for i in range(min(len(rel), 100)))
        expect_ndcg.append(dcg / idcg)
        hits = sorted(final_ranks)
        expect_ap.append(sum((i + 1) / float(r) for i, r in enumerate(hits)) / len(rel))
    buf = io.StringIO()
    T.write_run('sert', data, buf)
    run = T.parse_run(io.StringIO(buf.getvalue()))
    res = T.evaluate_run(run, qrels, k=100)
    assert abs(res['ndcg_cut_100'] - np.mean(expect_ndcg)) < 1e-9
    assert abs(res['map'] - np.mean(expect_ap)) < 1e-9


def test_written_rank_is_the_evaluated_rank_under_ties():
    """trec_eval re-sorts by score and breaks ties by document id, descending; write_run numbers the
    lines in that same order, so the rank column never disagrees with what gets evaluated."""
    data = {'7': [(0.5, 'B0001'), (0.5, 'B0003'), (0.9, 'B0002'), (0.5, 'B0002x')]}
    buf = io.StringIO()
    T.write_run('m', data, buf)
    lines = [l.split() for l in buf.getvalue().splitlines()]
    assert [l[2] for l in lines] == ['B0002', 'B0003', 'B0002x', 'B0001']
    assert [int(l[3]) for l in lines] == [1, 2, 3, 4]
    run = T.parse_run(io.StringIO(buf.getvalue()))
    assert T._ranked(run['7']) == [l[2] for l in lines]
