"""`Seq2SeqAttNN` — host-side mirror of the reference's model plug-in (model_combine.py:10-315).

main.py:65-66 loads the model as ``getattr(__import__(args.model), "Seq2SeqAttNN")(args_dict)`` and calls
``train(sess, item_dict, train_data, neighbor_dict, args, test_data, saver)`` / ``test(sess, test_data, args)``.
This class keeps those signatures (``sess`` / ``saver`` are accepted and ignored: there is no TensorFlow session),
the stdout lines of model_combine.py:201,229-230,244,247,255,288-293,310-314 and the NaN guard (:243-246), and
runs every step on the HIP engine (`tcar_amd.engine.TcarEngine`).  Differences that are deliberate:

  * the per-batch feed is int32 arrays from the tensorised sampler (host/sampler.py), not nested lists;
  * evaluation ranks / top-20 come from the `tcar_rank_topk` kernel, so the [B, N] score matrix never travels to
    the host (model_combine.py:293,296,301 argsort it twice per session on the CPU);
  * ILD / unexp are vectorised over an int category table (model_combine.py:174-194 are O(20^2) dict lookups);
  * dense weights are drawn from RandomState(2020) (TF's RNG stream of tf.set_random_seed(2020) cannot be
    reproduced); the embedding tables consume the GLOBAL numpy stream in creation order exactly like
    modules.py:32, so a seeded run draws the same negatives as the reference afterwards.
"""
from __future__ import annotations

import os
import time
from collections import OrderedDict

import numpy as np
import torch

from ..engine import TIME_NAMES, TIME_VOCAB, TcarEngine, VAR_ORDER
from . import metrics as M
from .sampler import Sampler


def _shapes(N, H, Ht):
    s = OrderedDict()
    s["item_emb"] = (N + 1, H)
    s["dec_pos"] = (40, H)
    for n, v in zip(TIME_NAMES, TIME_VOCAB):
        s[n] = (v, Ht)
    s["duration_embedding"] = (11, Ht)
    dense = {"multi_attention/input_linear_trans/w_3d": (2 * H, H), "multi_attention/cont_linear_trans/w_3d": (H, H),
             "multi_attention/inter_linear_trans/w_3d": (Ht, H), "multi_attention/res_linear_trans/w_3d": (H, 1),
             "multi_attention/query_trans1/w1": (2 * Ht, H), "multi_attention/query_trans1/b1": (H,),
             "multi_attention/query_trans2/w1": (H, 2 * H), "multi_attention/query_trans2/b1": (2 * H,),
             "attout_item_cont_trans/w1": (2 * H, 2 * H), "attout_item_cont_trans/b1": (2 * H,),
             "cont_attention/input_linear_trans/w_3d": (5 * Ht, H), "cont_attention/cont_linear_trans/w_3d": (H, H),
             "cont_attention/res_linear_trans/w_3d": (H, 1), "attout_pt_trans/w1": (5 * Ht, 5 * Ht),
             "attout_pt_trans/b1": (5 * Ht,)}
    for n in VAR_ORDER:
        if n in dense:
            s[n] = dense[n]
    return s


def initial_variables(N, H, Ht, emb_stddev, stddev, weight_seed=2020, lean=False):
    """Tables: np.random.normal on the global stream, creation order, row 0 zeroed when zero_pad
    (modules.py:32-34; dec_pos default stddev 0.02 and no pad, model_combine.py:57-64; duration no pad, :106).
    Dense weights ~ N(0, stddev) (modules.py:50-51,65) from a private stream.
    lean=True (multi-million-item catalogs): the item table is drawn in fp32 row chunks from a private generator (the
    fp64 draw of a 10M x 256 table alone is 20 GB)."""
    wr = np.random.RandomState(weight_seed)
    out = OrderedDict()
    for name, shp in _shapes(N, H, Ht).items():
        if name == "item_emb" and lean:
            g32 = np.random.default_rng(weight_seed + 1)
            t = np.empty(shp, dtype=np.float32)
            for lo in range(0, shp[0], 1 << 20):
                hi = min(shp[0], lo + (1 << 20))
                t[lo:hi] = g32.standard_normal((hi - lo, shp[1]), dtype=np.float32) * np.float32(emb_stddev)
            t[0] = 0.0
            out[name] = t
            continue
        if name == "dec_pos":
            t = np.random.normal(0, 0.02, shp)
        elif name in ("item_emb", "duration_embedding") or name in TIME_NAMES:
            t = np.random.normal(0, emb_stddev, shp)
            if name != "duration_embedding":
                t[0] = 0.0
        else:
            t = wr.normal(0, stddev, shp)
        out[name] = t.astype(np.float32)
    return out


def _idx0(sampler, i: int) -> int:
    """store row of the first example of batch i (its input length is the batch's T)"""
    return int(sampler.batch_indices(i)[0])


def prefetch_batches(sampler, depth: int = 4):
    """Iterate sampler.next_batch_arrays() from a producer thread (same order, same RNG stream: only this thread draws
    from numpy's global generator while it runs) so batch assembly overlaps the upload / step enqueue of the consumer;
    the ctypes step call and the H2D copy release the GIL."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    END = object()

    def produce():
        try:
            while sampler.has_next():
                q.put(sampler.next_batch_arrays())
            q.put(END)
        except BaseException as e:          # surfaced in the consumer
            q.put(e)

    th = threading.Thread(target=produce, daemon=True)
    th.start()
    while True:
        item = q.get()
        if item is END:
            break
        if isinstance(item, BaseException):
            raise item
        yield item
    th.join()


class Seq2SeqAttNN():
    """The memory network with context/temporal attention, MI355X edition."""

    def __init__(self, args):
        self.publish_time_MWDHM = np.asarray(args['publish_time_MWDHM'], dtype=np.int32)
        self.itemnum = args['itemnum']
        self.category_id = args['category_id']
        self.item_freq_dict_norm = args.get('item_freq_dict_norm')
        self.reverse_item = args['reverse_item']
        content = np.asarray(args['content_emb'], dtype=np.float32)
        self.candidate_n = content.shape[0]
        self.emb_stddev = args['emb_stddev']
        self.stddev = args['stddev']
        self.hidden_size = args['hidden_size']
        self.time_hidden_size = args['time_hidden_size']
        self.l2_emb = args.get('l2_emb', 0.0)
        self.batch_size = args['batch_size']
        self.epoch = args['epoch']
        self.neg_num = args['neg_num']
        self.gap_mode = args.get('gap_mode', 'active_t')
        self.neg_mode = args.get('neg_mode', 'uniform')
        self.neg_fast = bool(args.get('neg_fast', 0))
        self.device_sampler = bool(args.get('device_sampler', 0))   # form batches + draw negatives on the GPU
        self.seed = int(args.get('seed', 2020))
        self._ds_cache = {}
        self.curEpoch = 0
        self.error_during_train = False
        if content.shape[1] != self.hidden_size:
            raise ValueError("content_emb width %d != --hidden_size %d (model_combine.py:111 concatenates them)"
                             % (content.shape[1], self.hidden_size))
        N = self.candidate_n - 1
        print('size of seq_content', (None, None, self.hidden_size))
        print('size of seq_publish_t', (None, None, 5 * self.time_hidden_size))
        print('size of click_t', (None, 2 * self.time_hidden_size))
        params = args.get('initial_variables') or initial_variables(N, self.hidden_size, self.time_hidden_size,
                                                                    self.emb_stddev, self.stddev)
        self.variables_names = [n + ':0' for n in VAR_ORDER]
        print(self.variables_names)
        engine_cls = TcarEngine
        kw = {}
        if args.get('dp_group') is not None:
            from ..dp import DPEngine
            engine_cls, kw = DPEngine, {"group": args['dp_group']}
            if args.get('dp_mode', 'replica') == 'sharded':
                from ..sharded import ShardedEngine
                engine_cls = ShardedEngine
        self.engine = engine_cls(params, content, self.publish_time_MWDHM, lr=args['lr'], max_grad=args.get('max_grad'),
                                 device=args.get('device', 'cuda:0'), scoring=args.get('scoring', 'bf16x3-mixed'), **kw)
        self._cat = None
        self._store_cache = {}
        # data parallel (main.py --gpus N): every rank builds the SAME batches (same seeds) and keeps its contiguous shard
        self.dp_group = args.get('dp_group')
        if self.dp_group is not None:
            import torch.distributed as dist
            self.dp_rank, self.dp_world = dist.get_rank(self.dp_group), dist.get_world_size(self.dp_group)
        else:
            self.dp_rank, self.dp_world = 0, 1

    # ------------------------------------------------------------------------------------------ helpers
    def _category_table(self):
        if self._cat is None:
            self._cat = M.category_table(self.reverse_item, self.category_id, self.candidate_n - 1)
        return self._cat

    def printData(self, filename, batch_in, batch_out, batch_pred):
        os.makedirs('saved', exist_ok=True)
        with open('saved/CAR+P_Normal_predict_exa_' + filename + '.txt', 'a+') as f:
            for index in range(len(batch_in)):
                f.write('# batch in: {} # batch out: {} # batch pred: {} \n'.format(
                    str(batch_in[index]), str(batch_out[index]), str(batch_pred[index])))

    def getILD(self, recList):
        return float(M.ild_batch(np.asarray([recList]), self._category_table())[0])

    def getUnexp(self, inSeq, recList):
        if len(recList) == 0:
            return 0
        return float(M.unexp_batch(np.asarray([inSeq]), np.asarray([recList]), self._category_table())[0])

    def _sampler(self, data, neighbor_dict=None, item_dict=None, neg_num=None):
        # (len_dict, session_dict, session_time_dict) as util.py:56 returns it, or with a 4th element: a prebuilt
        # SessionStore whose integer example ids populate len_dict (large synthetic folds skip the dict form)
        len_d, sess_d, time_d = data[:3]
        store = data[3] if len(data) > 3 else None
        return Sampler(len_d, sess_d, time_d, neighbor_dict, item_dict, neg_num, batch_size=self.batch_size,
                       gap_mode=self.gap_mode, neg_mode=self.neg_mode, store=store, neg_fast=self.neg_fast)

    def _device_batches(self, data, sampler, neighbor_dict, item_dict, K):
        """Iterate C batch descriptors formed ON THE DEVICE (device_sampler.DeviceSampler): the host sampler has only done
        the bucketed shuffle (sampler.py:40-49); each batch costs one H2D copy of its example indices."""
        from ..device_sampler import DeviceSampler
        store = sampler.store
        mode = self.neg_mode if (neighbor_dict and K) else "uniform"
        key = (id(store), mode)
        ds = self._ds_cache.get(key)
        if ds is None:
            ds = self._ds_cache[key] = DeviceSampler(self.engine, store, mode, neighbor_dict if mode != "uniform" else None,
                                                     item_dict, seed=self.seed)
        k = K if (neighbor_dict and K) else 0
        if self.dp_world == 1:
            # the epoch's example indices go up ONCE; every feed is then formed from HBM-resident data, one batch ahead of the
            # step on a side stream (DeviceSampler.planned) — nothing crosses PCIe inside the epoch
            ds.plan([sampler.batch_indices(i) for i in range(sampler.batch_num)])
            for bt in ds.planned(k, self.gap_mode):
                yield bt, None, None
            return
        for i in range(sampler.batch_num):
            idx = sampler.batch_indices(i)
            if self.dp_world > 1:
                from ..dp import shard_bounds
                lo, hi, cap = shard_bounds(len(idx), self.dp_world, self.dp_rank)
                T = int(store.in_len[idx[0]])
                yield (ds.form(idx[lo:hi], k, self.gap_mode) if hi > lo else None), cap * T, idx[lo:hi]
            else:
                yield ds.form(idx, k, self.gap_mode), None, idx

    def _shard(self, feed):
        """this rank's contiguous shard of a batch (dp.shard_bounds) -> (sub-feed or None when empty, rows capacity / T)"""
        if self.dp_world == 1:
            return feed, None
        from ..dp import shard_bounds
        b = feed["seq"].shape[0]
        lo, hi, cap = shard_bounds(b, self.dp_world, self.dp_rank)
        if hi <= lo:
            return None, cap
        return {k: (v[lo:hi] if v is not None else None) for k, v in feed.items()}, cap

    def _allsum(self, values):
        """sum a list of python floats over the ranks"""
        if self.dp_world == 1:
            return values
        import torch.distributed as dist
        t = torch.tensor(values, dtype=torch.float64, device=self.engine.dev)
        dist.all_reduce(t, group=self.dp_group)
        return t.cpu().tolist()

    # -------------------------------------------------------------------------------------------- train
    def train(self, sess, item_dict, train_data, neighbor_dict, args, test_data=None, saver=None, threshold_acc=0.99):
        eng = self.engine
        for epoch in range(self.epoch):
            self.curEpoch = epoch
            print('Epoch {}'.format(epoch))
            batch = 0
            sampler = self._sampler(train_data, neighbor_dict, item_dict, args['neg_num'])
            # per-row fp64 accumulator of the epoch's losses: ONE small kernel per step (the reference sums python floats)
            acc = torch.zeros(max(int(args['batch_size']), 1), dtype=torch.float64, device=eng.dev)
            count = 0
            t0 = time.time()
            if self.device_sampler:
                for bt, cap_rows, _idx in self._device_batches(train_data, sampler, neighbor_dict, item_dict, args['neg_num']):
                    batch += 1
                    if self.dp_world > 1:
                        crt_loss = eng.train_step(None, bt=bt, cap_rows=cap_rows, T=int(sampler.store.in_len[_idx0(sampler, batch - 1)]),
                                                  K=args['neg_num'] if neighbor_dict else 0)
                    else:
                        crt_loss = eng.train_step(None, bt=bt, defer_update=True)   # applied inside the next step (flush below)
                    acc[:crt_loss.numel()].add_(crt_loss)
                    count += crt_loss.numel()
            for feed in (() if self.device_sampler else prefetch_batches(sampler)):
                batch += 1
                if batch < 3 and feed["neg"] is not None:
                    print(feed["neg"][0][:10].tolist())
                if self.dp_world > 1:
                    T = feed["seq"].shape[1]
                    sub, cap = self._shard(feed)
                    crt_loss = eng.train_step(sub, cap_rows=cap * T, T=T, K=(feed["neg"].shape[1] if feed["neg"] is not None else 0))
                else:
                    crt_loss = eng.train_step(feed, defer_update=True)   # [b] on device; no host sync inside the loop
                acc[:crt_loss.numel()].add_(crt_loss)
                count += crt_loss.numel()
            eng.flush()                                         # the last step's deferred update
            eng.check_forks()
            tot, cnt = self._allsum([float(acc.sum().item()), float(count)])
            avgc = tot / max(cnt, 1)
            self.train_seconds = time.time() - t0
            self.train_sessions = int(cnt)
            if np.isnan(avgc):
                print('Epoch {}: NaN error!'.format(str(epoch)))
                self.error_during_train = True
                return
            print('\tloss: {:.6f}'.format(avgc))
            if test_data is not None:
                recall = self.test(sess, test_data, args)
                if recall > threshold_acc:       # every rank takes part (the sharded export is a collective), rank 0 writes
                    modelname = self.save(args)
                    if self.dp_rank == 0:
                        print('Model saved - {}'.format(modelname))

    def save(self, args):
        suf = time.strftime("%Y%m%d%H%M", time.localtime()) + '-' + str(args.get('dataset', '')).replace('/', '_') \
            + '-' + str(args.get('split_way', '')).replace('/', '_') + '-' + str(args.get('foldnum', 0))
        path = os.path.join(args.get('modelpath', './ckpt/'), "model.ckpt-" + suf + ".npz")
        # plain arrays only (variables, Adam moments, beta powers, step): loadable without unpickling anything.  NOT
        # interchangeable with the reference's tf.train.Saver checkpoints (README.md).
        state = self.engine.export_state()          # catalog-sharded engine: all-gathers the owners' Adam moments
        if self.dp_rank == 0:                       # replicas are identical: one writer
            os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
            np.savez(path, **state)
        return path

    # --------------------------------------------------------------------------------------------- test
    def test(self, sess, test_data, args):
        print('Measuring...')
        eng = self.engine
        cat = self._category_table()
        hits, mrrs, ndcgs, ilds, unexps, losses = [], [], [], [], [], []
        sampler = self._sampler(test_data)
        batch = 0
        pending = []
        # ILD / unexp pair counts and the set of recommended items stay on the device (tcar_eval_diversity)
        if getattr(self, "_cat_on_engine", None) is not cat:
            eng.set_categories(cat)
            self._cat_on_engine = cat
        eng.reset_coverage()
        for feed in prefetch_batches(sampler):
            batch += 1
            feed, _cap = self._shard(feed)           # data parallel: every rank scores its shard of the batch
            if feed is None:
                continue
            bt = eng.upload(feed)
            rank, topk, ce = eng.eval_step(None, k=20, bt=bt)
            ild_c, unexp_c, n_rec = eng.eval_diversity(bt, topk)
            pending.append((feed, rank.clone(), topk.clone() if args.get('is_print') else None, ce.clone(), ild_c, unexp_c, n_rec))     # device results; drained below
            if batch < 3:
                tk = topk[0].cpu().numpy().tolist()
                print('batch_in:', feed["seq"][0].tolist())
                print('active_interval:', feed["gap"][0].tolist())
                print('input_click_week:', int(feed["cw"][0]))
                print('batch_out:', int(feed["label"][0]), args['publish_time'][int(feed["label"][0])])
                print('batch pred:', tk[:10])
        eng.check_forks()             # (synchronises) nothing below is reported from a run whose flag forks timed out
        for feed, rank, topk, ce, ild_c, unexp_c, n_rec in pending:
            r = rank.cpu().numpy()
            h, m, n = M.metrics_from_ranks(r, 20)
            hits += h.tolist()
            mrrs += m.tolist()
            ndcgs += n.tolist()
            losses += ce.cpu().numpy().tolist()
            il, un = M.diversity_from_counts(ild_c.cpu().numpy(), unexp_c.cpu().numpy(), n_rec.cpu().numpy(), feed["seq"].shape[1])
            ilds += il.tolist()
            unexps += un.tolist()
            if args.get('is_print'):
                self.printData(str(args['foldnum']) + '_' + str(self.curEpoch), feed["seq"].tolist(),
                               feed["label"].tolist(), topk.cpu().numpy().astype(np.int64).tolist())
        # sums over this rank's sessions, then over the ranks; coverage = union of the recommended items
        sums = self._allsum([float(np.sum(x)) for x in (losses, ilds, unexps, mrrs, hits, ndcgs)] + [float(len(hits))])
        n = max(sums[6], 1.0)
        if self.dp_world > 1:
            import torch.distributed as dist
            dist.all_reduce(eng._seen, op=dist.ReduceOp.MAX, group=self.dp_group)      # union of the ranks' byte maps
        n_covered = eng.coverage()
        m_loss, m_ild, m_unexp, m_mrr, m_hit, m_ndcg = [v / n for v in sums[:6]]
        print('avg loss...', m_loss)
        print('avg ILD...', m_ild)
        print('avg unexp...', m_unexp)
        print('len of result dict: ', n_covered)
        print('MRR@20: {}, Recall@20: {}, nDCG@20: {}'.format(m_mrr, m_hit, m_ndcg))
        self.last_metrics = {"mrr": m_mrr, "recall": m_hit, "ndcg": m_ndcg, "loss": m_loss, "ild": m_ild,
                             "unexp": m_unexp, "coverage": n_covered}
        return m_hit
