"""DPCGANSynthesizer: the reference's conditional tabular WGAN-GP trainer
(/root/reference/src/dp_cgans/synthesizers/dp_cgan.py:183-928) with the same constructor, ``fit``,
``sample``, ``set_device``, ``save``/``load`` and attributes, running every step on the gfx950
kernels through ``engine.StepEngine`` (no torch.nn / autograd on the path).
"""
import os
import warnings
from datetime import datetime

import numpy as np
import pandas as pd
import torch

from . import ops
from .data_sampler import DataSampler
from .data_transformer import DataTransformer
from .engine import GraphStep, StepEngine
from .parallel import Dist
from .rdp_accountant import compute_rdp, get_privacy_spent


def _derive_seed(base, label):
    """63-bit stream key for ``label``, one-way in ``base`` (blake2b)."""
    import hashlib
    h = hashlib.blake2b(int(base).to_bytes(16, 'little', signed=True) + label, digest_size=8).digest()
    return int.from_bytes(h, 'little') >> 1


class _Net(object):
    """What ``synth._generator`` / ``synth._discriminator`` expose: the reference module's
    ``state_dict`` keys and shapes over the engine's flat parameter buffers."""

    def __init__(self, getter, setter):
        self._get, self._set = getter, setter

    def state_dict(self):
        return self._get()

    def load_state_dict(self, sd, strict=True):
        have = self._get()
        if strict and set(sd.keys()) != set(have.keys()):
            raise RuntimeError('state_dict keys do not match: missing %s, unexpected %s'
                               % (sorted(set(have) - set(sd)), sorted(set(sd) - set(have))))
        self._set(sd)

    def parameters(self):
        return [v for k, v in self._get().items() if 'running' not in k and 'num_batches' not in k]


class DPCGANSynthesizer(object):
    """Conditional Table GAN Synthesizer (arguments as the reference, dp_cgan.py:190-239).

    Additional keyword arguments of this build: ``dp_mode`` ('reference': the reference's
    private=True, i.e. grad-hook noise + weight clamp; 'dpsgd': per-example clipping + Gaussian
    noise), ``dp_sigma``, ``dp_clip``, ``seed``, ``use_graph``, ``bn_sync`` (data parallel only: True -
    the generator's BatchNorm normalises over the global batch and the fit equals the single-process
    one; False - every rank over its own rows, as DistributedDataParallel does, 2 collectives per step
    instead of 6).

    ``seed=None`` (the default) keys every Philox stream - parameter init, batch draws and the DP
    noise - from OS entropy, as the reference draws from torch's / numpy's unseeded global generators
    (dp_cgan.py:581,645-646); rank 0's entropy is broadcast so data-parallel replicas stay identical.
    An explicit ``seed`` makes a fit reproducible and is a TEST HOOK: with ``private=True`` it also
    makes the Gaussian noise a publicly computable sequence, which voids the privacy guarantee (a
    warning says so).  The noise key is never written into a saved model.

    ``cuda``: the hot path exists for the GPU only.  ``cuda=False`` (which the reference honours by
    falling back to the CPU, dp_cgan.py:265-272) draws a one-time warning and runs on the GPU; a device
    string that does not name a GPU raises ``ValueError``.
    """

    def __init__(self, embedding_dim=128, generator_dim=(256, 256), discriminator_dim=(256, 256),
                 generator_lr=2e-4, generator_decay=1e-6, discriminator_lr=2e-4,
                 discriminator_decay=1e-6, batch_size=500, discriminator_steps=1,
                 log_frequency=True, verbose=False, epochs=300, pac=10, cuda=True, private=False,
                 wandb=False, ontology=None, saved_transformer=None, dp_mode='reference', dp_sigma=1.0,
                 dp_clip=1.0, seed=None, use_graph=True, bn_sync=True):
        assert batch_size % 2 == 0                      # dp_cgan.py:241

        self._embedding_dim = embedding_dim
        self._generator_dim = generator_dim
        self._discriminator_dim = discriminator_dim
        self._generator_lr = generator_lr
        self._generator_decay = generator_decay
        self._discriminator_lr = discriminator_lr
        self._discriminator_decay = discriminator_decay
        self._batch_size = batch_size
        self._discriminator_steps = discriminator_steps
        self._log_frequency = log_frequency
        self._verbose = verbose
        self._epochs = epochs
        self.pac = pac
        self.private = private
        self.ontology = ontology
        self.wandb = wandb
        self.saved_transformer = saved_transformer
        self.dp_mode, self.dp_sigma, self.dp_clip = dp_mode, dp_sigma, dp_clip
        self.seed = seed
        self.use_graph = use_graph
        self.bn_sync = bn_sync

        if private and seed is not None:
            warnings.warn('DPCGANSynthesizer(private=True, seed=%r): a fixed seed makes the DP noise a reproducible, '
                          'publicly computable sequence and voids the privacy guarantee; use it for tests only '
                          '(seed=None keys the noise from OS entropy)' % (seed,), UserWarning, stacklevel=2)
        self._device = self._resolve_device(cuda)

        self._transformer = None
        self._data_sampler = None
        self._generator = None
        self._discriminator = None
        self._engine = None
        self.loss_values = None

    _warned_no_cpu = False

    @classmethod
    def _resolve_device(cls, cuda):
        """The hot path exists for the GPU only: there is no CPU fallback (the reference falls back to
        'cpu' when ``cuda`` is false, dp_cgan.py:265-272).  ``cuda=False`` -> one warning per process and
        the GPU anyway; a device string that names no GPU -> ValueError."""
        if isinstance(cuda, (str, torch.device)):
            device = torch.device(cuda)
            if device.type != 'cuda':
                raise ValueError("dp_cgans_amd runs its training step on MI355X HIP kernels only: device %r is not a "
                                 "GPU and there is no CPU path (the reference's cuda=False / 'cpu' fallback, "
                                 "dp_cgan.py:265-272, has no counterpart here)" % (str(cuda),))
            if device.index is None:
                device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
            return device
        if not cuda and not cls._warned_no_cpu:
            DPCGANSynthesizer._warned_no_cpu = True
            warnings.warn('cuda=False requested, but dp_cgans_amd has no CPU path: the model runs on the GPU '
                          '(cuda:%s)' % os.environ.get('LOCAL_RANK', '0'), UserWarning, stacklevel=3)
        return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))

    # ------------------------------------------------------------------------------------------
    def _validate_discrete_columns(self, train_data, discrete_columns):
        """Check whether ``discrete_columns`` exists in ``train_data`` (dp_cgan.py:392-415)."""
        if isinstance(train_data, pd.DataFrame):
            invalid_columns = set(discrete_columns) - set(train_data.columns)
        elif isinstance(train_data, np.ndarray):
            invalid_columns = [c for c in discrete_columns if c < 0 or c >= train_data.shape[1]]
        else:
            raise TypeError('``train_data`` should be either pd.DataFrame or np.array.')
        if invalid_columns:
            raise ValueError('Invalid columns found: {}'.format(invalid_columns))

    def _require_gpu(self):
        if not torch.cuda.is_available():
            raise RuntimeError('dp_cgans_amd runs its training step on MI355X HIP kernels only; no GPU is visible '
                               'and there is no CPU fallback')

    def _base_seed(self, dist=None):
        """The key of this fit's Philox streams: the user's ``seed`` (test hook) or 63 bits of OS entropy,
        rank 0's for every rank.  Returned, never stored: a saved model does not carry the noise key."""
        if self.seed is not None:
            return int(self.seed)
        base = int.from_bytes(os.urandom(8), 'little') >> 1
        if dist is not None and dist.active:
            base = dist.broadcast_int(base, self._device)
        return base

    def _build_engine(self, span_info, dist=None, base_seed=None):
        world = 1 if dist is None else dist.world
        if self._batch_size % (world * self.pac) != 0:
            # every rank's shard must be whole pac groups (Discriminator.forward asserts B % pac == 0,
            # dp_cgan.py:67); dropping a remainder would leave the 1/B of the means wrong
            raise ValueError('batch_size %d is not a multiple of world size x pac = %d x %d'
                             % (self._batch_size, world, self.pac))
        if base_seed is None:
            base_seed = self._base_seed(dist)
        eng = StepEngine(span_info, self._embedding_dim, self._generator_dim, self._discriminator_dim,
                         self._batch_size // world,
                         self.pac, self._device, generator_lr=self._generator_lr, generator_decay=self._generator_decay,
                         discriminator_lr=self._discriminator_lr, discriminator_decay=self._discriminator_decay,
                         private=self.private, seed=base_seed, dist=dist, global_batch=self._batch_size,
                         dp_mode=self.dp_mode, dp_sigma=self.dp_sigma, dp_clip=self.dp_clip, bn_sync=self.bn_sync)
        self._engine = eng
        self._generator = _Net(eng.generator_state_dict, eng.load_generator_state_dict)
        self._discriminator = _Net(eng.discriminator_state_dict, eng.disc.load)
        return eng

    # ------------------------------------------------------------------------------------------
    def fit(self, train_data, discrete_columns=tuple(), epochs=None):
        """Fit the synthesizer (dp_cgan.py:420-835).

        Args:
            train_data (numpy.ndarray or pandas.DataFrame): training data, 2-dimensional.
            discrete_columns (list-like): discrete columns (names for a DataFrame, indices for an array).
        """
        self._require_gpu()
        self._validate_discrete_columns(train_data, discrete_columns)
        if epochs is None:
            epochs = self._epochs
        else:
            warnings.warn(
                ('`epochs` argument in `fit` method has been deprecated and will be removed '
                 'in a future version. Please pass `epochs` to the constructor instead'),
                DeprecationWarning)
        # every launch below goes to the current stream of self._device: make it the current device
        with torch.cuda.device(self._device):
            dist = Dist(device=self._device)    # first: it decides which rank writes files and whose entropy keys the fit
            try:
                self._fit(train_data, discrete_columns, epochs, dist)
                dist.barrier()
            finally:
                # a process group this fit created ends with it (one the caller initialised is left alone); the
                # engine then runs as a single rank: sample() draws and normalises per rank
                dist.shutdown()

    def _fit(self, train_data, discrete_columns, epochs, dist):
        base_seed = self._base_seed(dist)
        # the transformer is pickled with the model: its stream key is a one-way digest of the fit's key,
        # so a saved model never reveals the key of the DP noise stream
        self._transformer = DataTransformer(device=self._device, seed=_derive_seed(base_seed, b'transformer'))
        if self.saved_transformer and os.path.exists(self.saved_transformer):
            import joblib
            print('Found existing fitted transformer - ', self.saved_transformer)
            self._transformer = joblib.load(self.saved_transformer)
            self._transformer._device = self._device
        else:
            self._transformer.fit(train_data, discrete_columns)
            if dist.rank == 0:               # one writer: every rank shares the working directory
                try:
                    import joblib
                    joblib.dump(self._transformer, os.path.join(os.getcwd(), 'fitted_transformer.pkl'))   # dp_cgan.py:483
                except Exception as e:      # read-only cwd: training does not depend on the dump
                    warnings.warn('could not write fitted_transformer.pkl: %s' % e)

        table = self._transformer.transform_device(train_data)          # fp32 [N, Dd], stays in HBM
        n_rows = table.shape[0]
        span_info = self._transformer.span_info()
        self._data_sampler = DataSampler(table, span_info, self._log_frequency, device=self._device,
                                         seed=base_seed + 17 + dist.rank)
        eng = self._build_engine(span_info, dist, base_seed)
        eng.sampler = self._data_sampler

        self.loss_values = pd.DataFrame(columns=['Epoch', 'Generator Loss', 'Distriminator Loss'])
        steps_per_epoch = max(n_rows // self._batch_size, 1)
        if self._verbose:
            print('Training Configuration:')
            print(f'  Batch size: {self._batch_size}')
            print(f'  Steps per epoch: {steps_per_epoch}')
            print(f'  Generator dim: {self._generator_dim}')
            print(f'  Discriminator dim: {self._discriminator_dim}')
            print(f'  Device: {self._device}')
            print(f'  Data shape: {tuple(table.shape)}')

        gstep = GraphStep(eng, self._discriminator_steps) if self.use_graph else None
        rows = []
        filename = f'loss_output_{datetime.now().strftime("%Y%m%d_%H%M%S")}.txt'
        try:
            loss_file = open(filename, 'w') if dist.rank == 0 else None
        except OSError:
            loss_file = None
        if loss_file:
            loss_file.write('Epoch,Generator_Loss,Discriminator_Loss,Timestamp\n')
        self.epsilon = float('nan')
        import time
        torch.cuda.synchronize(self._device)
        t_loop = time.perf_counter()
        for i in range(epochs):
            for _ in range(steps_per_epoch):
                if gstep is not None:
                    gstep.run()
                else:
                    eng.train_step(self._discriminator_steps)
            losses = eng.global_losses()              # one sync per epoch, as the reference's .item() (dp_cgan.py:735-736)
            generator_loss = float(losses[2] + losses[3])
            discriminator_loss = float(losses[0])
            rows.append((i, generator_loss, discriminator_loss))
            if loss_file:
                timestamp = datetime.now().strftime('%Y-%m-%d %H:%M:%S')
                loss_file.write(f'{i},{generator_loss:.3f},{discriminator_loss:.3f},{timestamp}\n')
                loss_file.flush()
            if self.private:
                self.epsilon = self._privacy_report(i, steps_per_epoch, n_rows)
        torch.cuda.synchronize(self._device)
        # wall time of the step loop (the metric's denominator: rows/s = batch x steps / this), kept for reports
        self.step_loop_seconds = time.perf_counter() - t_loop
        self.steps_done = epochs * steps_per_epoch
        if loss_file:
            loss_file.close()
        self.loss_values = pd.DataFrame(rows, columns=['Epoch', 'Generator Loss', 'Distriminator Loss'])

    def _privacy_report(self, epoch, steps_per_epoch, n_rows):
        """RDP accounting per epoch (dp_cgan.py:808-824).  dp_mode 'reference' keeps the
        reference's call verbatim (sigma=1, steps = epoch index * steps per epoch, Q14); 'dpsgd'
        calls the same accountant with sigma = dp_sigma, q = batch / rows and one composition per
        discriminator update.

        The epsilon printed is a HEURISTIC in both modes, as the reference's is - not a proven bound:
        * the accountant assumes Poisson subsampling of records with probability q, but batches are
          drawn by the conditional sampler (with replacement, with data-dependent selection
          probabilities through the co-occurrence tables);
        * the clipped unit is a pac group of ``pac`` real rows, and one record can sit in several
          groups of a batch, so a record's contribution is bounded by m x dp_clip for its
          multiplicity m, not by dp_clip;
        * the generator update (and BatchNorm statistics) are not accounted.
        A faithful bound needs per-record sensitivity (rows drawn without replacement within a batch
        plus a group-privacy factor for pac); that is outside the reference's behaviour."""
        orders = [1 + x / 10. for x in range(1, 100)]
        delta = 2e-6
        if self.dp_mode == 'dpsgd':
            q = min(1.0, self._batch_size / n_rows)
            steps = (epoch + 1) * steps_per_epoch * self._discriminator_steps
            sigma = self.dp_sigma
        else:
            q = self._batch_size / n_rows
            steps = epoch * steps_per_epoch
            sigma = 1
        # compute_rdp is linear in `steps`: the 99 per-composition values are computed once per fit
        key = (min(q, 1.0), sigma)
        if getattr(self, '_rdp_cache', (None, None))[0] != key:
            self._rdp_cache = (key, compute_rdp(q=key[0], noise_multiplier=sigma, steps=1, orders=orders))
        rdp = self._rdp_cache[1] * steps
        epsilon, _, opt_order = get_privacy_spent(orders, rdp, target_delta=delta)
        if self._verbose:
            if epoch == 0:
                print('note: eps below is the reference\'s heuristic RDP estimate (Poisson-subsampling accountant '
                      'applied to conditionally sampled pac groups), not a proven bound.')
            print('differential privacy with eps = {:.3g} and delta = {}.'.format(epsilon, delta))
            print('The optimal RDP order is {}.'.format(opt_order))
            if opt_order == max(orders) or opt_order == min(orders):
                print('The privacy estimate is likely to be improved by expanding the set of orders.')
        return epsilon

    # ------------------------------------------------------------------------------------------
    def sample(self, n, condition_column=None, condition_value=None):
        """Sample data similar to the training data (dp_cgan.py:838-888).

        Choosing a condition_column and condition_value will increase the probability of the
        discrete condition_value happening in the condition_column.
        """
        self._require_gpu()
        with torch.cuda.device(self._device):
            return self._sample(n, condition_column, condition_value)

    def _sample(self, n, condition_column, condition_value):
        eng = self._engine
        if condition_column is not None and condition_value is not None:
            info = self._transformer.convert_column_name_value_to_id(condition_column, condition_value)
            global_condition_vec = self._data_sampler.generate_cond_from_condition_column_info(info, eng.B)
        else:
            global_condition_vec = None
        steps = n // eng.B + 1
        data = []
        for _ in range(steps):
            if global_condition_vec is not None:
                cond = torch.as_tensor(global_condition_vec).to(eng.dev)
            else:
                cond = self._data_sampler.sample_original_condvec_device(eng.B)
            data.append(eng.sample_forward(cond))
        data = torch.cat(data, dim=0)[:n]
        return self._transformer.inverse_transform_device(data)

    def set_device(self, device):
        """Move the model (dp_cgan.py:890-893).  A fitted model is rebuilt on the new GPU from its own
        state (networks, sampler table); a device that is not a GPU raises ValueError."""
        device = self._resolve_device(device)
        if self._engine is not None and device != self._device:
            st = self.__getstate__()
            st['_device'] = str(device)
            self.__setstate__(st)
        else:
            self._device = device

    def xai_discriminator(self, data_samples):
        """Discriminator scores [len(data_samples) / pac, 1] of that many real rows drawn for fresh
        conditional vectors (dp_cgan.py:896-928: the reference scores sampled rows, not the rows passed
        in; Dropout stays on, as there)."""
        eng = self._engine
        n_rows = len(pd.DataFrame(data_samples).T)
        if n_rows % eng.pac != 0:
            raise AssertionError('rows %% pac != 0')      # Discriminator.forward, dp_cgan.py:67
        with torch.cuda.device(self._device):
            return eng.score_real_rows(rows=n_rows)

    # ---- persistence (synthesizers/base.py:6-21) -----------------------------------------------
    def __getstate__(self):
        st = dict(self.__dict__)
        eng = st.pop('_engine', None)
        st.pop('_generator', None)
        st.pop('_discriminator', None)
        sampler = st.pop('_data_sampler', None)
        st['_device'] = str(self._device)
        if eng is not None:
            st['_saved'] = {
                'span_info': eng.span_info_np, 'generator': {k: v.cpu() for k, v in eng.generator_state_dict().items()},
                'discriminator': {k: v.cpu() for k, v in eng.discriminator_state_dict().items()},
                'table': sampler.table.cpu() if sampler is not None else None, 'batch': eng.B,
            }
        return st

    def __setstate__(self, st):
        saved = st.pop('_saved', None)
        self.__dict__.update(st)
        self._device = torch.device(self._device)
        self._engine = self._generator = self._discriminator = self._data_sampler = None
        self._saved = saved
        if saved is not None and torch.cuda.is_available():
            self._restore()

    def _restore(self):
        saved = self._saved
        if self._transformer is not None:
            self._transformer._device = self._device
        with torch.cuda.device(self._device):
            base_seed = self._base_seed()
            eng = self._build_engine(saved['span_info'], base_seed=base_seed)
            eng.load_generator_state_dict(saved['generator'])
            eng.disc.load(saved['discriminator'])
            if saved['table'] is not None:
                self._data_sampler = DataSampler(saved['table'], saved['span_info'], self._log_frequency,
                                                 device=self._device, seed=base_seed + 17)
                eng.sampler = self._data_sampler
        self._saved = None

    def save(self, path):
        torch.save(self, path)

    @classmethod
    def load(cls, path):
        model = torch.load(path, weights_only=False)      # a file this class wrote
        return model
