

def test_reference_checkpoint_unpickles_with_our_vocab_class():
    """tests/golden/vqvae_shipped_ckpt.bin was written by the reference (torch.save of args Namespace, a model.vocab.Vocab,
    gen_dict): it unpickles against scripts/model/vocab.py and rebuilds the as-shipped (GSSoft) model on the CPU side."""
    import os, sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    from model.vocab import Vocab
    from utils.train_utils import load_checkpoint_and_model
    path = os.path.join(root, "tests", "golden", "vqvae_shipped_ckpt.bin")
    args, net, _, lang, pose_dim = load_checkpoint_and_model(path, "cpu", "autoencoder_vq")
    assert isinstance(lang, Vocab) and lang.n_words == 13 and lang.index2word[Vocab.UNK_token] == "<UNK>"
    assert lang.word2count["the"] == 3 and lang.get_word_index("zebra") == Vocab.UNK_token
    assert lang.word_embedding_weights.shape == (13, 300)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert set(net.state_dict()) == set(raw["gen_dict"])
    for k, v in raw["gen_dict"].items():
        assert torch.equal(net.state_dict()[k], v), k
    assert args.autoencoder_vq_quantizer == "gssoft" and pose_dim == 40
    # our own Vocab behaves like the table the reference pickled
    v = Vocab("t")
    for w in "the quick brown fox jumps over the lazy dog the end".split():
        v.index_word(w)
    assert v.word2index == lang.word2index and v.word2count == lang.word2count and v.index2word == lang.index2word
    v.trim(2)
    assert v.n_words == 5 and v.word2index == {"the": 4}
