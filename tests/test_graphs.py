"""HIP-graph replay (dyglib_amd/graphs.py): a graphed TGN evaluation run is bit-identical to the eager run, including
the memory bank it leaves behind and the eager fallback on a ragged last batch."""
import numpy as np
import pytest

from tests import golden_cases as gc


@pytest.mark.gpu
def test_graphed_tgn_steps_equal_eager_steps():
    import torch
    from dyglib_amd import MemoryModel, MergeLayer, get_neighbor_sampler
    from dyglib_amd.graphs import GraphedStep
    c = gc.build_tgn_case("tgn_bip_l1_k10")
    cfg, d, dev = c["tgn_cfg"], c["data"], "cuda:0"

    def build():
        sampler = get_neighbor_sampler(d, "recent", seed=1, device=dev)
        m = MemoryModel(c["node_feat"], c["edge_feat"], sampler, time_feat_dim=cfg["time_feat_dim"], model_name="TGN", num_layers=cfg["num_layers"],
                        num_heads=cfg["num_heads"], dropout=0.1, device=dev)
        sd = m.state_dict(); sd.update({k: torch.from_numpy(v) for k, v in c["tgn_params"].items()}); m.load_state_dict(sd)
        mg = MergeLayer(172, 172, 172, 1); mg.load_state_dict({k: torch.from_numpy(v) for k, v in c["mparams"].items()})
        m, mg = m.to(dev).eval(), mg.to(dev).eval()
        m.memory_bank.__init_memory_bank__()
        k = cfg["num_neighbors"]

        def step(s, dd, ng, t, e):
            with torch.no_grad():
                a, b = m.compute_src_dst_node_temporal_embeddings(s, ng, t, edge_ids=None, edges_are_positive=False, num_neighbors=k)
                p, q = m.compute_src_dst_node_temporal_embeddings(s, dd, t, edge_ids=e, edges_are_positive=True, num_neighbors=k)
                return mg.link_probabilities(p, q), mg.link_probabilities(a, b), p, q
        return m, step

    B, nb = 40, 12
    rs = np.random.RandomState(3)
    uniq = np.unique(d.dst_node_ids)
    batches = []
    for i in range(nb):
        n = B if i < nb - 1 else 17                      # ragged last batch
        sl = slice(i * B, i * B + n)
        batches.append(tuple(torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in
                             (d.src_node_ids[sl], d.dst_node_ids[sl], uniq[rs.randint(0, len(uniq), n)], d.node_interact_times[sl], d.edge_ids[sl])))
    m_e, step_e = build()
    m_g, step_g = build()
    eager = [tuple(t.clone() for t in step_e(*b)) for b in batches]
    outs = [tuple(t.clone() for t in step_g(*b)) for b in batches[:2]]          # first steps eagerly (uploads, workspaces)
    graphed = GraphedStep(step_g, batches[2])
    for b in batches[2:]:
        outs.append(tuple(t.clone() for t in graphed(*b)))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(eager, outs)):
        for x, y in zip(a, b):
            assert torch.equal(x, y), i
    assert torch.equal(m_e.memory_bank.node_memories, m_g.memory_bank.node_memories)
    assert torch.equal(m_e.memory_bank.node_last_updated_times, m_g.memory_bank.node_last_updated_times)
