// GoldenDump.java — pins this repo's CPU oracle (and through it the HIP path) to the REAL reference.
//
// UNCOMPILED: there is no JDK in the build image or on the GPU box (tests/test_lib_loads.py records the probe), so this file
// has never been through javac.  It uses only the reference's public API as read from its sources:
//   ModelLoader.loadModel(Path, int, boolean, boolean)   src/main/java/org/beehive/gpullama3/model/loader/ModelLoader.java:113
//   Model.createNewState(), Model.forward(State, int, int), Model.configuration()      .../model/Model.java:37,61
//   State.logits / State.x / State.keyCache / State.valueCache (FloatTensor)             .../inference/state/State.java:31-45
//   FloatTensor.getFloat(int), FloatTensor.argmax()                                       .../tensor/standard/FloatTensor.java:71,138
// Model.forward is the pure-Java CPU path (InferenceCore.forwardJava :50, forwardJavaQwen3 :565, ...) when the model is loaded
// with useTornadovm = false — no TornadoVM device is touched, but the TornadoVM API jars must be on the class path because
// State imports their array types.
//
// What it does: loads a GGUF written by tools/reference_pin/make_pin_ggufs.py, feeds the java.util.Random(42) token stream
// LlamaBench uses (bench/LlamaBench.java:188-193) for nPrompt positions, continues greedily (FloatTensor.argmax) for nGreedy
// more, and writes every step's logits, the final x and the last position's K / V rows of every layer as little-endian f32.
// tests/test_reference_golden.py compares the C oracle and the HIP path with the dump bit for bit.
//
// Build + run (from the reference checkout, after `mvn -q package -DskipTests`), see INTEGRATION.md §6:
//   javac --enable-preview --release 21 --add-modules jdk.incubator.vector -cp target/classes:$TORNADO_API_JARS \
//         -d /tmp/pin /path/to/repo/tools/reference_pin/GoldenDump.java
//   java  --enable-preview --add-modules jdk.incubator.vector -cp /tmp/pin:target/classes:$TORNADO_API_JARS \
//         GoldenDump pin_ggufs/tiny_llama_q8_0.gguf /path/to/repo/tests/golden/reference/tiny_llama_q8_0.bin 6 24
// Modes: the dump records -Dllama.VectorBitSize (0 = scalar dots; 128 / 256 / 512 = the species; unset = the host's preferred species:
// 256 on AVX2, 512 on AVX-512 — every one of them has a counterpart in both oracles since round 5) and -Dllama.quantizeActivation
// (default true); the test picks the oracle mode — and the plan flags of the HIP path — from the header.
import java.io.DataOutputStream;
import java.io.FileOutputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.file.Path;
import java.util.Random;

import jdk.incubator.vector.VectorShape;

import org.beehive.gpullama3.inference.state.State;
import org.beehive.gpullama3.model.Configuration;
import org.beehive.gpullama3.model.Model;
import org.beehive.gpullama3.model.loader.ModelLoader;
import org.beehive.gpullama3.tensor.standard.FloatTensor;

public final class GoldenDump {
    private static void putFloats(DataOutputStream out, FloatTensor t, int off, int n) throws Exception {
        ByteBuffer b = ByteBuffer.allocate(4 * n).order(ByteOrder.LITTLE_ENDIAN);
        for (int i = 0; i < n; i++) b.putFloat(t.getFloat(off + i));
        out.write(b.array());
    }

    private static void putInts(DataOutputStream out, int... v) throws Exception {
        ByteBuffer b = ByteBuffer.allocate(4 * v.length).order(ByteOrder.LITTLE_ENDIAN);
        for (int x : v) b.putInt(x);
        out.write(b.array());
    }

    public static void main(String[] args) throws Exception {
        if (args.length < 4) {
            System.err.println("usage: GoldenDump model.gguf out.bin nPrompt nGreedy");
            System.exit(2);
        }
        int nPrompt = Integer.parseInt(args[2]), nGreedy = Integer.parseInt(args[3]);
        int steps = nPrompt + nGreedy - 1;
        Model model = ModelLoader.loadModel(Path.of(args[0]), steps + 8, true, false);      // loadWeights, CPU path
        Configuration c = model.configuration();
        State state = model.createNewState();
        int vocab = c.vocabularySize(), dim = c.dim(), layers = c.numberOfLayers(), kvDim = c.kvDim();
        // same expressions as FloatTensor.VECTOR_BIT_SIZE (:21) and Q8_0FloatTensor.QUANTIZE_ACTIVATION (:70)
        int vectorBits = Integer.getInteger("llama.VectorBitSize", VectorShape.preferredShape().vectorBitSize());
        boolean quantAct = Boolean.parseBoolean(System.getProperty("llama.quantizeActivation", "true"));

        int[] tokens = new int[steps + 1];
        Random rng = new Random(42);
        for (int i = 0; i < nPrompt; i++) tokens[i] = rng.nextInt(vocab);

        try (DataOutputStream out = new DataOutputStream(new FileOutputStream(args[1]))) {
            out.write("GL3REF01".getBytes("US-ASCII"));
            putInts(out, steps, vocab, dim, layers, kvDim, nPrompt, vectorBits, quantAct ? 1 : 0);
            java.io.ByteArrayOutputStream body = new java.io.ByteArrayOutputStream();
            DataOutputStream bo = new DataOutputStream(body);
            for (int pos = 0; pos < steps; pos++) {
                model.forward(state, tokens[pos], pos);
                putFloats(bo, state.logits, 0, vocab);
                if (pos >= nPrompt - 1) tokens[pos + 1] = state.logits.argmax();
            }
            putInts(out, tokens);
            out.write(body.toByteArray());
            putFloats(out, state.x, 0, dim);                                                   // after the final rmsnorm (in place)
            for (int l = 0; l < layers; l++) putFloats(out, state.keyCache[l], (steps - 1) * kvDim, kvDim);
            for (int l = 0; l < layers; l++) putFloats(out, state.valueCache[l], (steps - 1) * kvDim, kvDim);
        }
        System.out.println("GoldenDump: " + args[1] + " steps=" + steps + " vocab=" + vocab + " vectorBits=" + vectorBits + " quantizeActivation=" + quantAct
                + " java=" + System.getProperty("java.version") + " vm=" + System.getProperty("java.vm.name") + " arch=" + System.getProperty("os.arch"));
    }
}
