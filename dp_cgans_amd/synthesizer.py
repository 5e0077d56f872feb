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
    noise), ``dp_sigma``, ``dp_clip``, ``seed``, ``use_graph``.
    """

    def __init__(self, embedding_dim=128, generator_dim=(256, 256), discriminator_dim=(256, 256),
                 generator_lr=2e-4, generator_decay=1e-6, discriminator_lr=2e-4,
                 discriminator_decay=1e-6, batch_size=500, discriminator_steps=1,
                 log_frequency=True, verbose=False, epochs=300, pac=10, cuda=True, private=False,
                 wandb=False, ontology=None, saved_transformer=None, dp_mode='reference', dp_sigma=1.0,
                 dp_clip=1.0, seed=0, use_graph=True):
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

        # The hot path exists for the GPU only: there is no CPU fallback (the reference falls back
        # to 'cpu', dp_cgan.py:265-272).
        if isinstance(cuda, str):
            device = cuda
        else:
            device = 'cuda:%d' % int(os.environ.get('LOCAL_RANK', '0'))
        self._device = torch.device(device)

        self._transformer = None
        self._data_sampler = None
        self._generator = None
        self._discriminator = None
        self._engine = None
        self.loss_values = None

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

    def _build_engine(self, span_info, dist=None):
        eng = StepEngine(span_info, self._embedding_dim, self._generator_dim, self._discriminator_dim,
                         self._batch_size if dist is None or dist.world == 1 else self._batch_size // dist.world,
                         self.pac, self._device, generator_lr=self._generator_lr, generator_decay=self._generator_decay,
                         discriminator_lr=self._discriminator_lr, discriminator_decay=self._discriminator_decay,
                         private=self.private, seed=self.seed, dist=dist, global_batch=self._batch_size,
                         dp_mode=self.dp_mode, dp_sigma=self.dp_sigma, dp_clip=self.dp_clip)
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

        self._transformer = DataTransformer(device=self._device, seed=self.seed)
        if self.saved_transformer and os.path.exists(self.saved_transformer):
            import joblib
            print('Found existing fitted transformer - ', self.saved_transformer)
            self._transformer = joblib.load(self.saved_transformer)
            self._transformer._device = self._device
        else:
            self._transformer.fit(train_data, discrete_columns)
            try:
                import joblib
                joblib.dump(self._transformer, os.path.join(os.getcwd(), 'fitted_transformer.pkl'))   # dp_cgan.py:483
            except Exception as e:      # read-only cwd: training does not depend on the dump
                warnings.warn('could not write fitted_transformer.pkl: %s' % e)

        table = self._transformer.transform_device(train_data)          # fp32 [N, Dd], stays in HBM
        n_rows = table.shape[0]
        span_info = self._transformer.span_info()
        dist = Dist()
        self._data_sampler = DataSampler(table, span_info, self._log_frequency, device=self._device,
                                         seed=self.seed + 17 + dist.rank)
        eng = self._build_engine(span_info, dist)
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
        accounts the mechanism that actually ran: sigma = dp_sigma, one composition per
        discriminator update, examples = pac groups."""
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
        self._device = torch.device(device)

    def xai_discriminator(self, data_samples):
        """Discriminator score of real rows drawn for a fresh conditional vector (dp_cgan.py:896-928;
        Dropout stays on, as in the reference)."""
        eng = self._engine
        n_rows = len(pd.DataFrame(data_samples).T)
        if n_rows % eng.pac != 0:
            raise AssertionError('rows %% pac != 0')      # Discriminator.forward, dp_cgan.py:67
        return eng.score_real_rows()

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
        eng = self._build_engine(saved['span_info'])
        eng.load_generator_state_dict(saved['generator'])
        eng.disc.load(saved['discriminator'])
        if saved['table'] is not None:
            self._data_sampler = DataSampler(saved['table'], saved['span_info'], self._log_frequency,
                                             device=self._device, seed=self.seed + 17)
            eng.sampler = self._data_sampler
        self._saved = None

    def save(self, path):
        torch.save(self, path)

    @classmethod
    def load(cls, path):
        model = torch.load(path, weights_only=False)      # a file this class wrote
        return model
