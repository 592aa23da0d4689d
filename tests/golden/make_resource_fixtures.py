"""Cuts small DATA fixtures out of the reference's benchmark resources (run in the build
container only, where /root/reference exists):

    python tests/golden/make_resource_fixtures.py

resources/product-search/home_and_kitchen/{topics,qrel_test,qrel_validation,product_list} are
data files of the reference (topic strings, relevance judgements, entity ids), not source.  The
first 60 topics, every judgement line that belongs to them and the entities those lines name are
kept; tests/test_trec_fixtures_cpu.py parses and evaluates them.
"""
import os

SRC = '/root/reference/resources/product-search/home_and_kitchen'
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'product_search')
NUM_TOPICS = 60


def main():
    os.makedirs(DST, exist_ok=True)
    with open(os.path.join(SRC, 'topics')) as f:
        topics = [line for line in f][:NUM_TOPICS]
    ids = set(line.split(';', 1)[0] for line in topics)
    with open(os.path.join(DST, 'topics'), 'w') as f:
        f.writelines(topics)
    entities = set()
    for name in ('qrel_test', 'qrel_validation'):
        with open(os.path.join(SRC, name)) as f:
            lines = [line for line in f if line.split()[0] in ids]
        entities.update(line.split()[2] for line in lines)
        with open(os.path.join(DST, name), 'w') as f:
            f.writelines(lines)
    with open(os.path.join(SRC, 'product_list')) as f:
        products = [line for line in f if line.strip() in entities]
    with open(os.path.join(DST, 'product_list'), 'w') as f:
        f.writelines(products)
    print('%d topics, %d entities' % (len(topics), len(products)))


if __name__ == '__main__':
    main()
